/* TEST-ONLY declarations: the subset of TensorFlow's C++ API that tf_ops/hkv_ops_rocm.cc uses, so that the shim can be
 * syntax- and type-checked (`-fsyntax-only`) in an image without TensorFlow (tests/test_tf_shim.py).  Nothing here is
 * linked or shipped; signatures follow TensorFlow 2.16 (the version TFRA pins, R/README.md:109). */
#ifndef TFRA_STUB_TENSORFLOW_OP_KERNEL_H_
#define TFRA_STUB_TENSORFLOW_OP_KERNEL_H_
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace Eigen {
struct half { uint16_t x; };
struct GpuDevice { void* stream() const; };
}  // namespace Eigen

namespace tensorflow {
struct bfloat16 { uint16_t x; };
using tstring = std::string;

class Status {
 public:
  bool ok() const;
  void IgnoreError() const;
};
Status OkStatus();
namespace errors {
template <class... A> Status InvalidArgument(A...);
template <class... A> Status Internal(A...);
template <class... A> Status ResourceExhausted(A...);
}  // namespace errors
#define TF_RETURN_IF_ERROR(...)                    \
  do {                                             \
    ::tensorflow::Status _s = (__VA_ARGS__);       \
    if (!_s.ok()) return _s;                       \
  } while (0)

enum DataType { DT_INVALID = 0, DT_FLOAT = 1, DT_INT32 = 3, DT_UINT8 = 4, DT_INT8 = 6, DT_STRING = 7, DT_INT64 = 9, DT_BOOL = 10,
                DT_BFLOAT16 = 14, DT_HALF = 19, DT_RESOURCE = 20, DT_UINT64 = 23 };
using DataTypeVector = std::vector<DataType>;
std::string DataTypeString(DataType);
int DataTypeSize(DataType);

class TensorShape {
 public:
  TensorShape();
  TensorShape(std::initializer_list<int64_t>);
  int64_t dim_size(int) const;
  int dims() const;
  void AppendShape(const TensorShape&);
  std::string DebugString() const;
};
class PartialTensorShape {};
struct TensorShapeUtils {
  static bool IsVector(const TensorShape&);
  static bool IsScalar(const TensorShape&);
};
struct StringPiece { const char* data() const; size_t size() const; };
class ResourceHandle {};
class Tensor {
 public:
  const TensorShape& shape() const;
  int64_t NumElements() const;
  StringPiece tensor_data() const;
  template <class T> struct Scalar { T& operator()(); const T& operator()() const; };
  template <class T> Scalar<T> scalar();
  template <class T> Scalar<T> scalar() const;
};

class Allocator {
 public:
  virtual ~Allocator();
  virtual void* AllocateRaw(size_t alignment, size_t num_bytes) = 0;
  virtual void DeallocateRaw(void* ptr) = 0;
};
struct AllocatorAttributes { void set_on_host(bool); void set_gpu_compatible(bool); };
class DeviceBase { public: Allocator* GetAllocator(AllocatorAttributes); };
class Env {
 public:
  Status RecursivelyCreateDir(const std::string&);
  Status GetMatchingPaths(const std::string& pattern, std::vector<std::string>* results);
};
class NodeDef {};
template <class T> Status GetNodeAttr(const NodeDef&, const char* name, T* value);

using mutex = std::mutex;
using mutex_lock = std::lock_guard<std::mutex>;

class ResourceMgr;
class OpKernelConstruction {
 public:
  template <class T> Status GetAttr(const char* name, T* value) const;
  void CtxFailureWithWarning(const char*, int, const Status&);
};
class OpKernelContext {
 public:
  const Tensor& input(int);
  Status allocate_output(const char* name, const TensorShape&, Tensor** out);
  Status allocate_output(int index, const TensorShape&, Tensor** out, AllocatorAttributes);
  Status allocate_temp(DataType, const TensorShape&, Tensor* out);
  Status MatchSignature(const DataTypeVector& inputs, const DataTypeVector& outputs);
  template <class D> const D& eigen_device() const;
  DeviceBase* device() const;
  Env* env() const;
  ResourceMgr* resource_manager() const;
  const Status& status() const;
  void CtxFailureWithWarning(const char*, int, const Status&);
};
class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*);
  virtual ~OpKernel();
  virtual void Compute(OpKernelContext*) = 0;
  const NodeDef& def() const;
};
#define OP_REQUIRES_OK(CTX, ...)                                         \
  do {                                                                   \
    ::tensorflow::Status _s(__VA_ARGS__);                                \
    if (!_s.ok()) { (CTX)->CtxFailureWithWarning(__FILE__, __LINE__, _s); return; } \
  } while (0)
#define OP_REQUIRES(CTX, EXP, STATUS)                                    \
  do {                                                                   \
    if (!(EXP)) { (CTX)->CtxFailureWithWarning(__FILE__, __LINE__, (STATUS)); return; } \
  } while (0)

extern const char* const DEVICE_GPU;
namespace register_kernel {
struct Name {
  explicit Name(const char*);
  Name& Device(const char*);
  template <class T> Name& TypeConstraint(const char*);
  Name& HostMemory(const char*);
};
struct Registrar { Registrar(const Name&, std::function<OpKernel*(OpKernelConstruction*)>); };
}  // namespace register_kernel
#define TFRA_STUB_CAT2(a, b) a##b
#define TFRA_STUB_CAT(a, b) TFRA_STUB_CAT2(a, b)
#define REGISTER_KERNEL_BUILDER(BUILDER, ...)                                                              \
  static ::tensorflow::register_kernel::Registrar TFRA_STUB_CAT(tfra_stub_kernel_, __COUNTER__)(           \
      ::tensorflow::register_kernel::BUILDER,                                                              \
      [](::tensorflow::OpKernelConstruction* c) -> ::tensorflow::OpKernel* { return new __VA_ARGS__(c); })
}  // namespace tensorflow
#endif
