/* TEST-ONLY declarations, see op_kernel.h in this directory. */
#ifndef TFRA_STUB_TENSORFLOW_SHAPE_INFERENCE_H_
#define TFRA_STUB_TENSORFLOW_SHAPE_INFERENCE_H_
#include "tensorflow/core/framework/op_kernel.h"
namespace tensorflow {
namespace shape_inference {
struct ShapeHandle {};
struct DimensionHandle {};
struct ShapeAndType {
  ShapeAndType(ShapeHandle s, DataType t);
  ShapeHandle shape;
  DataType dtype;
};
class InferenceContext {
 public:
  ShapeHandle input(int);
  void set_output(int, ShapeHandle);
  Status WithRank(ShapeHandle, int64_t, ShapeHandle*);
  Status WithRankAtLeast(ShapeHandle, int64_t, ShapeHandle*);
  Status Merge(DimensionHandle, DimensionHandle, DimensionHandle*);
  Status Concatenate(ShapeHandle, ShapeHandle, ShapeHandle*);
  Status Subshape(ShapeHandle, int64_t start, int64_t end, ShapeHandle*);
  Status MakeShapeFromPartialTensorShape(const PartialTensorShape&, ShapeHandle*);
  ShapeHandle UnknownShape();
  ShapeHandle UnknownShapeOfRank(int64_t);
  ShapeHandle Scalar();
  DimensionHandle Dim(ShapeHandle, int64_t);
  bool RankKnown(ShapeHandle);
  int32_t Rank(ShapeHandle);
  template <class T> Status GetAttr(const char* name, T* value);
  const std::vector<ShapeAndType>* input_handle_shapes_and_types(int);
  void set_output_handle_shapes_and_types(int, const std::vector<ShapeAndType>&);
};
}  // namespace shape_inference
}  // namespace tensorflow
#endif
