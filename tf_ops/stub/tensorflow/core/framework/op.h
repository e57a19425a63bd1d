/* TEST-ONLY declarations, see op_kernel.h in this directory. */
#ifndef TFRA_STUB_TENSORFLOW_OP_H_
#define TFRA_STUB_TENSORFLOW_OP_H_
#include "tensorflow/core/framework/shape_inference.h"
namespace tensorflow {
namespace register_op {
struct OpDefBuilderWrapper {
  explicit OpDefBuilderWrapper(const char* name);
  OpDefBuilderWrapper& Input(const char*);
  OpDefBuilderWrapper& Output(const char*);
  OpDefBuilderWrapper& Attr(const char*);
  OpDefBuilderWrapper& SetIsStateful();
  OpDefBuilderWrapper& SetShapeFn(std::function<Status(shape_inference::InferenceContext*)>);
};
}  // namespace register_op
#define REGISTER_OP(NAME) \
  static ::tensorflow::register_op::OpDefBuilderWrapper& TFRA_STUB_CAT(tfra_stub_op_, __COUNTER__) = \
      ::tensorflow::register_op::OpDefBuilderWrapper(NAME)
}  // namespace tensorflow
#endif
