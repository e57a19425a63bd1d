/* TEST-ONLY declarations, see framework/op_kernel.h. */
#ifndef TFRA_STUB_TENSORFLOW_ENV_VAR_H_
#define TFRA_STUB_TENSORFLOW_ENV_VAR_H_
#include "tensorflow/core/framework/op_kernel.h"
namespace tensorflow {
Status ReadStringFromEnvVar(const std::string& name, const std::string& default_val, std::string* value);
Status ReadInt64FromEnvVar(const std::string& name, int64_t default_val, int64_t* value);   // tensorflow/core/util/env_var.h
}
#endif
