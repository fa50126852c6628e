// Drop-in include path of the reference (include/heyoka/logging.hpp:19-24: set_logger_level_trace() ... _critical()):
// forwards to the stage logger of the MI355X-native implementation (heyoka_amd/csrc/logging.hpp).
#pragma once
#include "../../heyoka_amd/csrc/logging.hpp"

#ifndef HEYOKA_AMD_NAMESPACE_ALIAS
#define HEYOKA_AMD_NAMESPACE_ALIAS
namespace heyoka = heyoka_amd;
#endif
