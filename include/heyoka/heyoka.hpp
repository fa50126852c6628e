// Drop-in include path of the reference (include/heyoka/heyoka.hpp): forwards to the MI355X-native implementation under
// heyoka_amd/csrc/ and exposes it as namespace heyoka, so that sources written against the reference's headers
// compile unchanged with -I <repo>/include -lheyoka_amd.
#pragma once
#include "../../heyoka_amd/csrc/taylor_adaptive_batch.hpp"
#include "../../heyoka_amd/csrc/ensemble.hpp"
#include "../../heyoka_amd/csrc/model.hpp"
#include "../../heyoka_amd/csrc/cfunc.hpp"
#include "../../heyoka_amd/csrc/logging.hpp"
#include "math/constants.hpp"
#include "math/kepDE.hpp"
#include "math/kepF.hpp"

#ifndef HEYOKA_AMD_NAMESPACE_ALIAS
#define HEYOKA_AMD_NAMESPACE_ALIAS
namespace heyoka = heyoka_amd;
#endif
