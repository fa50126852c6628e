// Double-length (hi, lo) floating-point arithmetic for the time coordinate
// (reference: include/heyoka/detail/dfloat.hpp:36-270). Plain IEEE operations: this header must be
// compiled without value-changing optimisations (-ffast-math is never used in this project).
#pragma once

#include <cmath>
#include <utility>

namespace heyoka_amd::detail
{

struct dfloat {
    double hi = 0, lo = 0;

    dfloat() = default;
    explicit dfloat(double x) : hi(x), lo(0) {}
    dfloat(double h, double l) : hi(h), lo(l) {}

    explicit operator double() const
    {
        return hi;
    }
};

inline bool isfinite(const dfloat &x)
{
    return std::isfinite(x.hi) && std::isfinite(x.lo);
}

// Knuth's error-free transformation of a sum (no magnitude requirement).
inline std::pair<double, double> eft_add_knuth(double a, double b)
{
    const double x = a + b;
    const double z = x - a;
    const double y = (a - (x - z)) + (b - z);
    return {x, y};
}

// Dekker's error-free transformation (requires |a| >= |b|).
inline std::pair<double, double> eft_add_dekker(double a, double b)
{
    const double x = a + b;
    const double y = (a - x) + b;
    return {x, y};
}

inline dfloat operator+(const dfloat &a, const dfloat &b)
{
    const auto [x_hi, y_hi] = eft_add_knuth(a.hi, b.hi);
    const auto [x_lo, y_lo] = eft_add_knuth(a.lo, b.lo);
    auto [u, v] = eft_add_dekker(x_hi, y_hi + x_lo);
    const auto [u2, v2] = eft_add_dekker(u, v + y_lo);
    return {u2, v2};
}

inline dfloat operator-(const dfloat &x, const dfloat &y)
{
    return x + dfloat(-y.hi, -y.lo);
}

inline dfloat operator+(const dfloat &x, double y)
{
    return x + dfloat(y);
}

inline dfloat operator-(const dfloat &x, double y)
{
    return x - dfloat(y);
}

inline bool operator<(const dfloat &x, const dfloat &y)
{
    return (x.hi < y.hi) || (x.hi == y.hi && x.lo < y.lo);
}

inline bool operator>(const dfloat &x, const dfloat &y)
{
    return (x.hi > y.hi) || (x.hi == y.hi && x.lo > y.lo);
}

inline bool operator>=(const dfloat &x, const dfloat &y)
{
    return (x.hi > y.hi) || (x.hi == y.hi && x.lo >= y.lo);
}

inline bool operator<=(const dfloat &x, const dfloat &y)
{
    return (x.hi < y.hi) || (x.hi == y.hi && x.lo <= y.lo);
}

inline bool operator==(const dfloat &x, const dfloat &y)
{
    return x.hi == y.hi && x.lo == y.lo;
}

} // namespace heyoka_amd::detail
