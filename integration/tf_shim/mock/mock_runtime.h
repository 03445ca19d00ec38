// Driver-side API of the mock runtime (mock_driver.cc): run a registered op by name.  See op_kernel.h.
#pragma once
#include <string>
#include <vector>
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"

namespace tensorflow {
namespace mock {
struct OpDef {
  std::string name;
  std::vector<std::string> input_names, output_names;
  std::vector<DataType> input_types, output_types;
  AttrMap attr_defaults;                       // attrs WITH a default
  std::map<std::string, AttrValue::Kind> attr_kinds;
  std::vector<std::string> host_memory;        // input names the kernel registration pins to host memory
  KernelFactory make;
};
const OpDef* FindOp(const std::string& name);
std::vector<std::string> RegisteredOps();
// Checks the invocation against the registration (input count, dtypes, host / device placement, attr names and kinds; attrs
// without a default must be given), constructs the kernel, runs Compute() on `stream`, returns its outputs.
Status RunOp(const std::string& name, const std::vector<Tensor>& inputs, const AttrMap& attrs, void* stream,
             std::vector<Tensor>* outputs);
}  // namespace mock
}  // namespace tensorflow
