/* TEST-ONLY declarations, see op_kernel.h in this directory. */
#ifndef TFRA_STUB_TENSORFLOW_RESOURCE_MGR_H_
#define TFRA_STUB_TENSORFLOW_RESOURCE_MGR_H_
#include "tensorflow/core/framework/op_kernel.h"
namespace tensorflow {
namespace core {
class RefCounted { public: void Ref() const; bool Unref() const; virtual ~RefCounted(); };
class ScopedUnref { public: explicit ScopedUnref(const RefCounted*); };
template <class T> class RefCountPtr { public: void reset(T* p = nullptr); T* get() const; };
}  // namespace core
class ResourceBase : public core::RefCounted { public: virtual std::string DebugString() const; virtual int64_t MemoryUsed() const; };
class ResourceMgr {
 public:
  template <class T> Status LookupOrCreate(const std::string& container, const std::string& name, T** resource,
                                            std::function<Status(T**)> creator);
  template <class T> Status Delete(const std::string& container, const std::string& name);
};
class ContainerInfo {
 public:
  Status Init(ResourceMgr*, const NodeDef&, bool use_node_name_as_default);
  ResourceMgr* resource_manager() const;
  const std::string& container() const;
  const std::string& name() const;
  bool resource_is_private_to_kernel() const;
};
const ResourceHandle& HandleFromInput(OpKernelContext*, int input);
template <class T> Status LookupResource(OpKernelContext*, const ResourceHandle&, T** value);
template <class T> ResourceHandle MakeResourceHandle(OpKernelContext*, const std::string& container, const std::string& name);
}  // namespace tensorflow
#endif
