// MOCK RUNTIME (see op_kernel.h in this directory): REGISTER_OP records the op's inputs / outputs / attrs (with their
// defaults) in a registry the driver checks every invocation against.
#pragma once
#include <string>
#include <vector>
namespace tensorflow {
struct OpDefBuilderMock {
  explicit OpDefBuilderMock(const char* n) : name(n) {}
  std::string name;
  std::vector<std::string> inputs, outputs, attrs;       // the spec strings as written: "x: float", "act: int = 0"
  OpDefBuilderMock& Input(const char* s) { inputs.push_back(s); return *this; }
  OpDefBuilderMock& Output(const char* s) { outputs.push_back(s); return *this; }
  OpDefBuilderMock& Attr(const char* s) { attrs.push_back(s); return *this; }
  template <typename F> OpDefBuilderMock& SetShapeFn(F) { return *this; }
};
struct OpRegistrar {
  OpRegistrar(const OpDefBuilderMock& def);               // mock_runtime.cc  (implicit on purpose: REGISTER_OP(...).Input(...)...)
};
}  // namespace tensorflow
#define TFGX_MOCK_OPCAT_(a, b) a##b
#define TFGX_MOCK_OPCAT(a, b) TFGX_MOCK_OPCAT_(a, b)
#define REGISTER_OP(NAME) static ::tensorflow::OpRegistrar TFGX_MOCK_OPCAT(op_reg_, __COUNTER__) = ::tensorflow::OpDefBuilderMock(NAME)
