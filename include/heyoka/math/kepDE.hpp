// Drop-in include path of the reference (include/heyoka/math/kepDE.hpp): forwards to the MI355X-native implementation under
// heyoka_amd/csrc/ and exposes it as namespace heyoka, so that sources written against the reference's headers
// compile unchanged with -I <repo>/include -lheyoka_amd. (kepDE is defined through the registry of node rules.)
#pragma once
#include "../../../heyoka_amd/csrc/node_rule.hpp"

#ifndef HEYOKA_AMD_NAMESPACE_ALIAS
#define HEYOKA_AMD_NAMESPACE_ALIAS
namespace heyoka = heyoka_amd;
#endif
