// Drop-in include path of the reference (include/heyoka/math/constants.hpp): heyoka::pi, a function without arguments
// (reference: constants.hpp:117) - here a registered node rule (heyoka_amd/csrc/builtin_rules.cpp).
#pragma once
#include "../../../heyoka_amd/csrc/node_rule.hpp"

namespace heyoka_amd
{
// NOTE: defined in the user's translation units (the library is fully initialised by then).
inline const expression pi = pi_constant();
} // namespace heyoka_amd

#ifndef HEYOKA_AMD_NAMESPACE_ALIAS
#define HEYOKA_AMD_NAMESPACE_ALIAS
namespace heyoka = heyoka_amd;
#endif
