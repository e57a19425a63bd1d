/* TEST-ONLY declarations, see op_kernel.h in this directory. */
#ifndef TFRA_STUB_TENSORFLOW_LOOKUP_INTERFACE_H_
#define TFRA_STUB_TENSORFLOW_LOOKUP_INTERFACE_H_
#include "tensorflow/core/framework/resource_mgr.h"
namespace tensorflow {
namespace lookup {
class LookupInterface : public ResourceBase {
 public:
  virtual size_t size() const = 0;
  virtual Status Find(OpKernelContext*, const Tensor& keys, Tensor* values, const Tensor& default_value) = 0;
  virtual Status Insert(OpKernelContext*, const Tensor& keys, const Tensor& values) = 0;
  virtual Status Remove(OpKernelContext*, const Tensor& keys) = 0;
  virtual Status ExportValues(OpKernelContext*) = 0;
  virtual Status ImportValues(OpKernelContext*, const Tensor& keys, const Tensor& values) = 0;
  virtual DataType key_dtype() const = 0;
  virtual DataType value_dtype() const = 0;
  virtual TensorShape key_shape() const = 0;
  virtual TensorShape value_shape() const = 0;
  virtual Status CheckKeyAndValueTensorsForInsert(const Tensor& keys, const Tensor& values);
  virtual Status CheckKeyAndValueTensorsForImport(const Tensor& keys, const Tensor& values);
  virtual Status CheckKeyTensorForRemove(const Tensor& keys);
  int64_t MemoryUsed() const override;
};
}  // namespace lookup
}  // namespace tensorflow
#endif
