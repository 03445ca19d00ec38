// MOCK of the TensorFlow headers tfgx_tf_ops.cc includes — just enough declarations for `g++ -fsyntax-only`
// (tests/test_abi.py::test_tf_shim_compiles_against_mock_headers).  TensorFlow is not installable in this image; the
// mock keeps the shim type-checked against the SHAPE of the real API (OpKernel / OpKernelContext / Tensor /
// REGISTER_OP / OP_REQUIRES ...).  It is not TensorFlow and nothing links against it.
#pragma once
#include <cstdint>
#include <initializer_list>
#include <string>

namespace tensorflow {
typedef std::int32_t int32;
typedef std::uint8_t uint8;
enum DataType { DT_FLOAT = 1, DT_INT32 = 3, DT_UINT8 = 4 };
extern const char* const DEVICE_GPU;
extern const char* const DEVICE_CPU;

class Status {
 public:
  Status() {}
  explicit Status(const std::string& m) : msg_(m), ok_(false) {}
  bool ok() const { return ok_; }
  std::string msg_;
  bool ok_ = true;
};
namespace errors {
inline Status InvalidArgument(const std::string& m) { return Status(m); }
inline Status Internal(const std::string& m) { return Status(m); }
}  // namespace errors

class TensorShape {
 public:
  TensorShape() {}
  TensorShape(std::initializer_list<std::int64_t>) {}
};

template <typename T>
struct Flat {
  T* data() const { return nullptr; }
};

class Tensor {
 public:
  int dims() const { return 0; }
  std::int64_t dim_size(int) const { return 0; }
  std::int64_t NumElements() const { return 0; }
  template <typename T> Flat<T> flat() { return Flat<T>(); }
  template <typename T> Flat<const T> flat() const { return Flat<const T>(); }
};

struct GpuDeviceMock {
  void* stream() const { return nullptr; }
};

class OpKernelConstruction {
 public:
  template <typename T> Status GetAttr(const char*, T*) { return Status(); }
  void CtxFailure(const Status&) {}
  void CtxFailureWithWarning(const Status&) {}
};

class OpKernelContext {
 public:
  const Tensor& input(int) { return t_; }
  Status allocate_output(int, const TensorShape&, Tensor**) { return Status(); }
  Status allocate_temp(DataType, const TensorShape&, Tensor*) { return Status(); }
  const GpuDeviceMock& eigen_gpu_device() const { return d_; }
  void CtxFailure(const Status&) {}
  void CtxFailureWithWarning(const Status&) {}
 private:
  Tensor t_;
  GpuDeviceMock d_;
};

class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*) {}
  virtual ~OpKernel() {}
  virtual void Compute(OpKernelContext* ctx) = 0;
};

struct KernelDefBuilderMock {
  KernelDefBuilderMock& Device(const char*) { return *this; }
  KernelDefBuilderMock& HostMemory(const char*) { return *this; }
};
inline KernelDefBuilderMock Name(const char*) { return KernelDefBuilderMock(); }
}  // namespace tensorflow

#define OP_REQUIRES(CTX, EXP, STATUS)        \
  do {                                       \
    if (!(EXP)) {                            \
      (CTX)->CtxFailure((STATUS));           \
      return;                                \
    }                                        \
  } while (0)
#define OP_REQUIRES_OK(CTX, ...)             \
  do {                                       \
    ::tensorflow::Status s_(__VA_ARGS__);    \
    if (!s_.ok()) {                          \
      (CTX)->CtxFailureWithWarning(s_);      \
      return;                                \
    }                                        \
  } while (0)
#define TFGX_MOCK_CAT_(a, b) a##b
#define TFGX_MOCK_CAT(a, b) TFGX_MOCK_CAT_(a, b)
#define REGISTER_KERNEL_BUILDER(BUILDER, ...) \
  static ::tensorflow::KernelDefBuilderMock TFGX_MOCK_CAT(kernel_reg_, __COUNTER__) = (BUILDER)
