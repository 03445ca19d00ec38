// MOCK (see op_kernel.h in this directory).
#pragma once
namespace tensorflow {
namespace shape_inference {
class InferenceContext {};
}  // namespace shape_inference
}  // namespace tensorflow
