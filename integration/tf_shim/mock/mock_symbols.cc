// definitions for the two extern strings of the mock (only needed if someone links the mock; the test is -fsyntax-only)
#include "tensorflow/core/framework/op_kernel.h"
namespace tensorflow {
const char* const DEVICE_GPU = "GPU";
const char* const DEVICE_CPU = "CPU";
}  // namespace tensorflow
