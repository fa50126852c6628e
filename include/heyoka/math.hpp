// Drop-in include path of the reference (include/heyoka/math.hpp): forwards to the MI355X-native implementation under
// heyoka_amd/csrc/ and exposes it as namespace heyoka, so that sources written against the reference's headers
// compile unchanged with -I <repo>/include -lheyoka_amd.
#pragma once
#include <heyoka/math/acos.hpp>
#include <heyoka/math/acosh.hpp>
#include <heyoka/math/asin.hpp>
#include <heyoka/math/asinh.hpp>
#include <heyoka/math/atan.hpp>
#include <heyoka/math/atan2.hpp>
#include <heyoka/math/atanh.hpp>
#include <heyoka/math/constants.hpp>
#include <heyoka/math/cos.hpp>
#include <heyoka/math/cosh.hpp>
#include <heyoka/math/erf.hpp>
#include <heyoka/math/exp.hpp>
#include <heyoka/math/kepDE.hpp>
#include <heyoka/math/kepE.hpp>
#include <heyoka/math/kepF.hpp>
#include <heyoka/math/log.hpp>
#include <heyoka/math/logical.hpp>
#include <heyoka/math/pow.hpp>
#include <heyoka/math/prod.hpp>
#include <heyoka/math/relational.hpp>
#include <heyoka/math/relu.hpp>
#include <heyoka/math/select.hpp>
#include <heyoka/math/sigmoid.hpp>
#include <heyoka/math/sin.hpp>
#include <heyoka/math/sinh.hpp>
#include <heyoka/math/sqrt.hpp>
#include <heyoka/math/sum.hpp>
#include <heyoka/math/tan.hpp>
#include <heyoka/math/tanh.hpp>
#include <heyoka/math/time.hpp>


#ifndef HEYOKA_AMD_NAMESPACE_ALIAS
#define HEYOKA_AMD_NAMESPACE_ALIAS
namespace heyoka = heyoka_amd;
#endif
