// MOCK RUNTIME of the TensorFlow headers tfgx_tf_ops.cc includes.  TensorFlow is not installable in this image, so the
// shim is (1) type-checked against these declarations (`g++ -fsyntax-only`, tests/test_abi.py) and (2) LINKED AND EXECUTED
// against them (round 4): Tensor owns real device / host memory, OpKernelContext hands out inputs and allocates outputs,
// REGISTER_OP / REGISTER_KERNEL_BUILDER fill registries, and mock_driver.cc runs ops by name — so every Compute() body of
// the shim (shape checks, argument marshalling into the tfgx C ABI) runs on the GPU (tests/test_gpu_tf_shim.py).
// It has the SHAPE of the real API (OpKernel / OpKernelContext / Tensor / REGISTER_OP / OP_REQUIRES ...) and nothing else:
// it is not TensorFlow.  No HIP header is included here (the syntax check uses plain g++): memory comes from the four
// functions declared below, defined in mock_runtime.cc.
#pragma once
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace tensorflow {
typedef std::int32_t int32;
typedef std::uint8_t uint8;
enum DataType { DT_INVALID = 0, DT_FLOAT = 1, DT_INT32 = 3, DT_UINT8 = 4, DT_INT64 = 9 };
extern const char* const DEVICE_GPU;
extern const char* const DEVICE_CPU;

namespace mock {
void* DeviceAlloc(std::size_t bytes);      // hipMalloc (mock_runtime.cc)
void DeviceFree(void* p);
void* HostAlloc(std::size_t bytes);
void HostFree(void* p);
std::size_t SizeOf(DataType t);
}  // namespace mock

class Status {
 public:
  Status() {}
  explicit Status(const std::string& m) : msg_(m), ok_(false) {}
  bool ok() const { return ok_; }
  const std::string& message() const { return msg_; }
  std::string msg_;
  bool ok_ = true;
};
namespace errors {
inline Status InvalidArgument(const std::string& m) { return Status("InvalidArgument: " + m); }
inline Status Internal(const std::string& m) { return Status("Internal: " + m); }
}  // namespace errors

class TensorShape {
 public:
  TensorShape() {}
  TensorShape(std::initializer_list<std::int64_t> d) : dims_(d) {}
  explicit TensorShape(const std::vector<std::int64_t>& d) : dims_(d) {}
  int dims() const { return static_cast<int>(dims_.size()); }
  std::int64_t dim_size(int i) const { return dims_[static_cast<std::size_t>(i)]; }
  std::int64_t num_elements() const {
    std::int64_t n = 1;
    for (std::int64_t d : dims_) n *= d;
    return n;
  }
  std::vector<std::int64_t> dims_;
};

template <typename T>
struct Flat {
  T* p;
  T* data() const { return p; }
};

class Tensor {
 public:
  Tensor() {}
  // memory: device (hipMalloc) unless `host`; always at least one element, so data() is never NULL for an empty tensor
  Tensor(DataType t, const TensorShape& s, bool host = false) : dtype_(t), shape_(s), host_(host) {
    const std::size_t bytes = static_cast<std::size_t>(s.num_elements() > 0 ? s.num_elements() : 1) * mock::SizeOf(t);
    void* p = host ? mock::HostAlloc(bytes) : mock::DeviceAlloc(bytes);
    buf_ = std::shared_ptr<void>(p, host ? mock::HostFree : mock::DeviceFree);
  }
  int dims() const { return shape_.dims(); }
  std::int64_t dim_size(int i) const { return shape_.dim_size(i); }
  std::int64_t NumElements() const { return shape_.num_elements(); }
  DataType dtype() const { return dtype_; }
  const TensorShape& shape() const { return shape_; }
  bool in_host_memory() const { return host_; }
  void* raw() const { return buf_.get(); }
  template <typename T> Flat<T> flat() { return Flat<T>{static_cast<T*>(buf_.get())}; }
  template <typename T> Flat<const T> flat() const { return Flat<const T>{static_cast<const T*>(buf_.get())}; }

 private:
  DataType dtype_ = DT_INVALID;
  TensorShape shape_;
  bool host_ = false;
  std::shared_ptr<void> buf_;
};

struct GpuDeviceMock {
  void* stream_ = nullptr;
  void* stream() const { return stream_; }
};

struct AttrValue {
  enum Kind { kInt, kBool, kFloat } kind = kInt;
  std::int64_t i = 0;
  bool b = false;
  float f = 0.0f;
};
typedef std::map<std::string, AttrValue> AttrMap;

class OpKernelConstruction {
 public:
  explicit OpKernelConstruction(const AttrMap& a) : attrs_(a) {}
  Status GetAttr(const char* name, int* v) { return Get(name, AttrValue::kInt, [&](const AttrValue& a) { *v = static_cast<int>(a.i); }); }
  Status GetAttr(const char* name, std::int64_t* v) { return Get(name, AttrValue::kInt, [&](const AttrValue& a) { *v = a.i; }); }
  Status GetAttr(const char* name, bool* v) { return Get(name, AttrValue::kBool, [&](const AttrValue& a) { *v = a.b; }); }
  Status GetAttr(const char* name, float* v) { return Get(name, AttrValue::kFloat, [&](const AttrValue& a) { *v = a.f; }); }
  void CtxFailure(const Status& s) { status_ = s; }
  void CtxFailureWithWarning(const Status& s) { status_ = s; }
  const Status& status() const { return status_; }

 private:
  template <typename F> Status Get(const char* name, AttrValue::Kind kind, F set) {
    auto it = attrs_.find(name);
    if (it == attrs_.end()) return Status(std::string("no attr named ") + name);
    if (it->second.kind != kind) return Status(std::string("attr ") + name + " has another type");
    set(it->second);
    return Status();
  }
  AttrMap attrs_;
  Status status_;
};

class OpKernelContext {
 public:
  OpKernelContext(const std::vector<Tensor>& inputs, const std::vector<DataType>& output_types, void* stream)
      : inputs_(inputs), output_types_(output_types), outputs_(output_types.size()) { d_.stream_ = stream; }
  const Tensor& input(int i) { return inputs_.at(static_cast<std::size_t>(i)); }
  Status allocate_output(int i, const TensorShape& s, Tensor** t) {
    if (i < 0 || static_cast<std::size_t>(i) >= outputs_.size()) return Status("allocate_output: no such output");
    outputs_[static_cast<std::size_t>(i)] = Tensor(output_types_[static_cast<std::size_t>(i)], s);
    *t = &outputs_[static_cast<std::size_t>(i)];
    return Status();
  }
  Status allocate_temp(DataType t, const TensorShape& s, Tensor* out) {
    *out = Tensor(t, s);
    return Status();
  }
  const GpuDeviceMock& eigen_gpu_device() const { return d_; }
  void CtxFailure(const Status& s) { status_ = s; }
  void CtxFailureWithWarning(const Status& s) { status_ = s; }
  const Status& status() const { return status_; }
  std::vector<Tensor>& outputs() { return outputs_; }

 private:
  std::vector<Tensor> inputs_;
  std::vector<DataType> output_types_;
  std::vector<Tensor> outputs_;
  GpuDeviceMock d_;
  Status status_;
};

class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*) {}
  virtual ~OpKernel() {}
  virtual void Compute(OpKernelContext* ctx) = 0;
};

struct KernelDefBuilderMock {
  std::string name, device;
  std::vector<std::string> host_memory;
  KernelDefBuilderMock& Device(const char* d) { device = d; return *this; }
  KernelDefBuilderMock& HostMemory(const char* a) { host_memory.push_back(a); return *this; }
};
inline KernelDefBuilderMock Name(const char* n) {
  KernelDefBuilderMock b;
  b.name = n;
  return b;
}
typedef std::function<OpKernel*(OpKernelConstruction*)> KernelFactory;
struct KernelRegistrar {
  KernelRegistrar(const KernelDefBuilderMock& def, KernelFactory make);    // mock_runtime.cc
};
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
#define REGISTER_KERNEL_BUILDER(BUILDER, ...)                                                        \
  static ::tensorflow::KernelRegistrar TFGX_MOCK_CAT(kernel_reg_, __COUNTER__)(                      \
      (BUILDER), [](::tensorflow::OpKernelConstruction* c) -> ::tensorflow::OpKernel* { return new __VA_ARGS__(c); })
