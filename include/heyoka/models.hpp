// Drop-in include path of the reference (include/heyoka/models.hpp): forwards to the MI355X-native implementation under
// heyoka_amd/csrc/ and exposes it as namespace heyoka, so that sources written against the reference's headers
// compile unchanged with -I <repo>/include -lheyoka_amd.
#pragma once
#include <heyoka/model/cr3bp.hpp>
#include <heyoka/model/fixed_centres.hpp>
#include <heyoka/model/mascon.hpp>
#include <heyoka/model/nbody.hpp>
#include <heyoka/model/pendulum.hpp>
#include <heyoka/model/rotating.hpp>


#ifndef HEYOKA_AMD_NAMESPACE_ALIAS
#define HEYOKA_AMD_NAMESPACE_ALIAS
namespace heyoka = heyoka_amd;
#endif
