// MOCK (see op_kernel.h in this directory).
#pragma once
namespace tensorflow {
struct OpDefBuilderMock {
  OpDefBuilderMock& Input(const char*) { return *this; }
  OpDefBuilderMock& Output(const char*) { return *this; }
  OpDefBuilderMock& Attr(const char*) { return *this; }
  template <typename F> OpDefBuilderMock& SetShapeFn(F) { return *this; }
};
}  // namespace tensorflow
#define TFGX_MOCK_OPCAT_(a, b) a##b
#define TFGX_MOCK_OPCAT(a, b) TFGX_MOCK_OPCAT_(a, b)
#define REGISTER_OP(NAME) static ::tensorflow::OpDefBuilderMock TFGX_MOCK_OPCAT(op_reg_, __COUNTER__) = ::tensorflow::OpDefBuilderMock()
