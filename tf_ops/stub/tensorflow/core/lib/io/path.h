/* TEST-ONLY declarations, see framework/op_kernel.h. */
#ifndef TFRA_STUB_TENSORFLOW_PATH_H_
#define TFRA_STUB_TENSORFLOW_PATH_H_
#include <string>
namespace tensorflow { namespace io { std::string JoinPath(const std::string&, const std::string&); } }
#endif
