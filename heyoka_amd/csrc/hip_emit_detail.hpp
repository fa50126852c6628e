// Internal pieces shared by the HIP code generators (hip_emit.cpp, hip_emit_cluster.cpp).
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <functional>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hip_emit.hpp"
#include "node_rule.hpp"

namespace heyoka_amd::emit_detail
{

// Common device-side prelude: argument block, double-length arithmetic, helpers.
extern const char *const prelude;

// The device code of the node rules a program uses (node_rule.hpp), to be emitted right behind the prelude.
inline std::string rules_source(const taylor_program &p)
{
    std::vector<std::uint32_t> ids;
    for (const auto &n : p.nodes) {
        if (n.kind == func_kind::custom) {
            ids.push_back(n.rule);
        }
    }
    return node_rules_device_source(ids);
}

// Synchronisation of the LDS exchange between the lanes of ONE wavefront (a system never spans
// wavefronts). The LDS pipeline executes the DS instructions of a wave in issue order, so a ds_write by
// one lane is visible to a later ds_read of any lane of the same wave without any hardware wait: all
// that is needed is that the *compiler* neither reorders LDS accesses across the point nor keeps slab
// values cached in registers. A memory fence here (even at wavefront scope) makes the compiler drain
// *all* outstanding memory operations (s_waitcnt vmcnt(0)), including the global jet-scratch stores,
// 40-80 times per step - measured: 72 % of the wave cycles parked in s_waitcnt.
inline constexpr const char *wsync_macro
    = "#define HY_WSYNC() do { asm volatile(\"\" ::: \"memory\"); __builtin_amdgcn_wave_barrier(); "
      "asm volatile(\"\" ::: \"memory\"); } while (0)\n";

// Scaling + safety factor of the step-size selector, folded on the host in double precision.
double rhofac(std::uint32_t order);

// Dense-output kernel.
void emit_dout(std::ostringstream &os, const taylor_program &p, const emit_options &opts);

struct ssa_emitter {
    const taylor_program &p;
    std::uint32_t order;
    std::ostringstream os;
    std::uint64_t counter = 0;
    std::uint64_t n_stmt = 0;
    // vals[k * n_u + u]: name (or literal) of the order-k coefficient of u variable u.
    std::vector<std::string> vals;

    // Constant u variables (constant_uvars(), decompose.hpp).
    std::vector<char> cu;

    ssa_emitter(const taylor_program &prog, std::uint32_t ord)
        : p(prog), order(ord), vals(static_cast<std::size_t>(prog.n_u) * (ord + 1u)), cu(constant_uvars(prog))
    {
    }

    std::string &val(std::uint32_t u, std::uint32_t k)
    {
        return vals[static_cast<std::size_t>(k) * p.n_u + u];
    }

    std::string def(const std::string &expr)
    {
        const auto name = "t" + std::to_string(counter++);
        os << "const double " << name << " = " << expr << ";\n";
        ++n_stmt;
        return name;
    }

    // Pairwise reduction (reference: src/detail/llvm_helpers_algo.cpp:271-308).
    std::string pairwise(std::vector<std::string> v, const char *op)
    {
        assert(!v.empty());
        while (v.size() != 1u) {
            std::vector<std::string> nv;
            for (std::size_t i = 0; i < v.size(); i += 2u) {
                if (i + 1u == v.size()) {
                    nv.push_back(v[i]);
                } else {
                    nv.push_back(def(v[i] + " " + op + " " + v[i + 1u]));
                }
            }
            v.swap(nv);
        }
        return v[0];
    }

    std::string pairwise_sum(std::vector<std::string> v)
    {
        return pairwise(std::move(v), "+");
    }

    // Sum of the terms of a CONVOLUTION (the products of a prod / sum_sq / pow / div / sin ... rule, in loop order). The
    // reference's default mode adds them pairwise (src/math/prod.cpp:386-395); its compact mode accumulates them in a
    // running sum which starts from 0 (src/math/prod.cpp:686-698). The running sum is the cheaper of the two in
    // straight-line code: with FMA contraction it is one instruction per term (k + 1 FMAs) where the pairwise tree needs
    // ~1.5 (k + 1) (half of the products cannot be fused into an addition). emit_options::sum_order selects
    // (hip_emit.hpp); the sums over the ARGUMENTS of sum() / sum_sq() are pairwise in both modes.
    bool running_sums = false;
    // sum_sq(): one running sum over the half convolutions of ALL the arguments (not an order of the additions of the
    // reference: only with the automatic sum_order of the straight-line generator and not under kw::exact_division).
    bool merge_sum_sq = false;
    std::string conv_sum(std::vector<std::string> v)
    {
        if (!running_sums) {
            return pairwise(std::move(v), "+");
        }
        auto acc = v.at(0);
        for (std::size_t j = 1; j < v.size(); ++j) {
            acc = def(acc + " + " + v[j]);
        }
        return acc;
    }

    // Optional replacement names for specific numerical operands (keyed by address of the operand in
    // the program): used by the cluster generator for constants that differ between isomorphic clusters.
    std::map<const operand *, std::string> numpar_override;

    std::string numpar(const operand &o) const
    {
        assert(o.type != operand::kind::uvar);
        if (const auto it = numpar_override.find(&o); it != numpar_override.end()) {
            return it->second;
        }
        if (o.type == operand::kind::num) {
            return fp_literal(o.value);
        }
        return "par_" + std::to_string(o.idx);
    }

    static bool is_var(const operand &o)
    {
        return o.type == operand::kind::uvar;
    }

    static std::string mul(const std::string &a, const std::string &b)
    {
        return a + " * " + b;
    }

    // A coefficient which is the literal +0 (state variables and u variables whose derivatives vanish identically from some
    // order on: x' = number, x' = v with a constant v). NOTE: not (-0x0p+0).
    static bool is_zero_lit(const std::string &s)
    {
        return s == "0.0" || s == "0x0p+0";
    }
    // Folding of those literals in the generator (the compiler does the same on the products; what it cannot remove without
    // fast-math flags are the chains built on top: 0 * h + 0 is a NaN for a non-finite h): x - (+0) = x for every x, signed
    // zeros and NaNs included; (+0) * RN(1 / d) = +0. HEYOKA_AMD_UNROLLED_TRIM=0 switches it off (A/B).
    bool fold_zeros = false;
    // Products whose factor is a scaled copy of another u variable (prod(number, u): the masses and couplings of a model):
    // the convolution runs over u and the number multiplies its sum, within 1 ulp of the reference's order of the
    // operations - not under kw::exact_division. Returns u (and the number, "-" for -1) or the variable itself.
    bool fold_scaled = false;
    std::uint32_t scaled_parent(std::uint32_t u, std::string &c) const
    {
        if (u < p.n_eq) {
            return u;
        }
        const auto &n = p.nodes[u - p.n_eq];
        if (n.kind != func_kind::prod || n.args.size() != 2u || is_var(n.args[0]) == is_var(n.args[1])) {
            return u;
        }
        const auto &cv = is_var(n.args[0]) ? n.args[1] : n.args[0];
        const auto &v = is_var(n.args[0]) ? n.args[0] : n.args[1];
        if (numpar_override.count(&cv) != 0u) {
            return u;
        }
        c = (cv.type == operand::kind::num && cv.value == -1.) ? std::string("-") : numpar(cv);
        return v.idx;
    }

    // Exponentiation by squaring (reference: pow_ebs(), src/math/pow.cpp:136-152).
    std::string pow_ebs(const std::string &base, std::uint32_t e)
    {
        if (e == 0u) {
            return "1.0";
        }
        if (e == 1u) {
            return base;
        }
        const auto sq = def(mul(base, base));
        if (e % 2u == 0u) {
            return pow_ebs(sq, e / 2u);
        }
        const auto tmp = pow_ebs(sq, (e - 1u) / 2u);
        return def(mul(base, tmp));
    }

    // Order-0 evaluation of pow (reference: get_pow_eval_algo(), src/math/pow.cpp:292-355).
    std::string pow_eval(const std::string &b, double ex)
    {
        if (std::isfinite(ex) && ex == std::trunc(ex) && std::abs(ex) <= 16.) {
            if (ex >= 0) {
                return pow_ebs(b, static_cast<std::uint32_t>(ex));
            }
            const auto tmp = pow_ebs(b, static_cast<std::uint32_t>(-ex));
            return def("1.0 / " + tmp);
        }
        if (std::isfinite(ex) && ex != std::trunc(ex)) {
            const auto y = 2 * ex;
            if (y == std::trunc(y) && std::abs(y) <= 16.) {
                const auto sq = def("sqrt(" + b + ")");
                if (y >= 0) {
                    return pow_ebs(sq, static_cast<std::uint32_t>(y));
                }
                const auto tmp = pow_ebs(sq, static_cast<std::uint32_t>(-y));
                return def("1.0 / " + tmp);
            }
        }
        return def("hy_pow(" + b + ", " + fp_literal(ex) + ")");
    }

    // Emit the order-k coefficient of node i.
    void node(std::uint32_t i, std::uint32_t k)
    {
        const auto &n = p.nodes[i];
        const auto u = p.n_eq + i;
        auto &out = val(u, k);
        const auto &a = n.args;

        switch (n.kind) {
            case func_kind::num_identity:
                out = (k == 0u) ? numpar(a[0]) : "0.0";
                break;
            case func_kind::time:
                // Reference: src/math/time.cpp:81-101.
                out = (k == 0u) ? "t_hi" : (k == 1u ? "1.0" : "0.0");
                break;
            case func_kind::sum: {
                // Reference: src/math/sum.cpp:185-235.
                std::vector<std::string> terms;
                for (const auto &o : a) {
                    if (is_var(o)) {
                        terms.push_back(val(o.idx, k));
                    } else {
                        terms.push_back(k == 0u ? numpar(o) : "0.0");
                    }
                }
                if (fold_zeros && k > 0u) {
                    // x + (+0) = x (up to the sign of a zero result): the coefficients of order >= 1 of u + number ARE those
                    // of u - no second history for the compiler to keep (x + 0.0 is not foldable without fast-math flags).
                    std::vector<std::string> nz;
                    for (auto &t : terms) {
                        if (!is_zero_lit(t)) {
                            nz.push_back(std::move(t));
                        }
                    }
                    if (nz.empty()) {
                        out = "0.0";
                        break;
                    }
                    terms.swap(nz);
                }
                out = pairwise_sum(std::move(terms));
                break;
            }
            case func_kind::sub: {
                // Reference: src/detail/sub.cpp:60-124.
                if (is_var(a[0]) && is_var(a[1])) {
                    if (fold_zeros && is_zero_lit(val(a[1].idx, k))) {
                        out = val(a[0].idx, k);
                        break;
                    }
                    out = def(val(a[0].idx, k) + " - " + val(a[1].idx, k));
                } else if (is_var(a[0])) {
                    out = (k == 0u) ? def(val(a[0].idx, 0) + " - " + numpar(a[1])) : val(a[0].idx, k);
                } else if (is_var(a[1])) {
                    out = (k == 0u) ? def(numpar(a[0]) + " - " + val(a[1].idx, 0)) : def("-" + val(a[1].idx, k));
                } else {
                    out = (k == 0u) ? def(numpar(a[0]) + " - " + numpar(a[1])) : "0.0";
                }
                break;
            }
            case func_kind::prod: {
                // Reference: src/math/prod.cpp:314-395.
                if (a.size() != 2u) {
                    throw std::invalid_argument("The Taylor derivative of a product can be computed only for "
                                                "products of 2 terms");
                }
                if (is_var(a[0]) && is_var(a[1]) && (cu[a[0].idx] != 0 || cu[a[1].idx] != 0)) {
                    // A constant factor (all its coefficients beyond order 0 are zero): the convolution reduces to
                    // c^[0] * x^[k] - the other k products of the reference's formula are exact zeros.
                    if (cu[a[0].idx] != 0 && cu[a[1].idx] != 0) {
                        out = (k == 0u) ? def(mul(val(a[0].idx, 0), val(a[1].idx, 0))) : "0.0";
                    } else {
                        const auto &c = (cu[a[0].idx] != 0) ? a[0] : a[1];
                        const auto &v = (cu[a[0].idx] != 0) ? a[1] : a[0];
                        out = def(mul(val(c.idx, 0), val(v.idx, k)));
                    }
                } else if (is_var(a[0]) && is_var(a[1])) {
                    // (fold_scaled: a factor which is itself number * u is read as u, the number multiplies the sum - the
                    // scaled copy of u's coefficient history is never kept.)
                    std::string c0, c1;
                    const auto i0 = fold_scaled ? scaled_parent(a[0].idx, c0) : a[0].idx;
                    const auto i1 = fold_scaled ? scaled_parent(a[1].idx, c1) : a[1].idx;
                    std::vector<std::string> terms;
                    for (std::uint32_t j = 0; j <= k; ++j) {
                        terms.push_back(def(mul(val(i0, k - j), val(i1, j))));
                    }
                    out = conv_sum(std::move(terms));
                    for (const auto *c : {&c0, &c1}) {
                        if (*c == "-") {
                            out = def("-" + out);
                        } else if (!c->empty()) {
                            out = def(mul(*c, out));
                        }
                    }
                } else if (!is_var(a[0]) && !is_var(a[1])) {
                    if (k != 0u) {
                        out = "0.0";
                    } else if (a[0].type == operand::kind::num && a[0].value == -1.) {
                        out = def("-" + numpar(a[1]));
                    } else {
                        out = def(mul(numpar(a[0]), numpar(a[1])));
                    }
                } else {
                    const auto &v = is_var(a[0]) ? a[0] : a[1];
                    const auto &c = is_var(a[0]) ? a[1] : a[0];
                    if (&c == &a[0] && c.type == operand::kind::num && c.value == -1. && numpar_override.count(&c) == 0u) {
                        out = def("-" + val(v.idx, k));
                    } else {
                        out = def(mul(numpar(c), val(v.idx, k)));
                    }
                }
                break;
            }
            case func_kind::div: {
                // Reference: src/detail/div.cpp:62-160.
                if (is_var(a[1])) {
                    if (k == 0u) {
                        const auto num = is_var(a[0]) ? val(a[0].idx, 0) : numpar(a[0]);
                        out = def(num + " / " + val(a[1].idx, 0));
                    } else {
                        std::vector<std::string> terms;
                        for (std::uint32_t j = 1; j <= k; ++j) {
                            terms.push_back(def(mul(val(u, k - j), val(a[1].idx, j))));
                        }
                        const auto acc = conv_sum(std::move(terms));
                        if (is_var(a[0])) {
                            out = def("(" + val(a[0].idx, k) + " - " + acc + ") / " + val(a[1].idx, 0));
                        } else {
                            out = def("(-" + acc + ") / " + val(a[1].idx, 0));
                        }
                    }
                } else if (is_var(a[0])) {
                    out = def(val(a[0].idx, k) + " / " + numpar(a[1]));
                } else {
                    out = (k == 0u) ? def(numpar(a[0]) + " / " + numpar(a[1])) : "0.0";
                }
                break;
            }
            case func_kind::sum_sq: {
                // Reference: src/detail/sum_sq.cpp:100-245.
                std::vector<std::string> tmp;
                if (merge_sum_sq && k > 0u && std::all_of(a.begin(), a.end(), [](const operand &o) { return is_var(o); })) {
                    // ONE accumulator for the half sums of all the arguments (like the pair kernels): the doubling and the
                    // additions between the arguments are paid once per order instead of once per argument.
                    std::string acc;
                    const auto jend = (k % 2u == 1u) ? (k - 1u) / 2u + 1u : k / 2u;
                    for (const auto &o : a) {
                        for (std::uint32_t j = 0; j < jend; ++j) {
                            acc = chain(acc, val(o.idx, k - j), val(o.idx, j));
                        }
                    }
                    out = def(acc + " + " + acc);
                    if (k % 2u == 0u) {
                        for (const auto &o : a) {
                            const auto &hv = val(o.idx, k / 2u);
                            out = def(mul(hv, hv) + " + " + out);
                        }
                    }
                    break;
                }
                if (k % 2u == 1u) {
                    for (const auto &o : a) {
                        if (is_var(o)) {
                            std::vector<std::string> terms;
                            for (std::uint32_t j = 0; j <= (k - 1u) / 2u; ++j) {
                                terms.push_back(def(mul(val(o.idx, k - j), val(o.idx, j))));
                            }
                            tmp.push_back(conv_sum(std::move(terms)));
                        } else {
                            tmp.emplace_back("0.0");
                        }
                    }
                    const auto s = pairwise_sum(std::move(tmp));
                    out = def(s + " + " + s);
                } else {
                    for (const auto &o : a) {
                        if (is_var(o)) {
                            const auto &hv = val(o.idx, k / 2u);
                            auto sq = def(mul(hv, hv));
                            if (k > 0u) {
                                std::vector<std::string> terms;
                                for (std::uint32_t j = 0; j <= (k - 2u) / 2u; ++j) {
                                    terms.push_back(def(mul(val(o.idx, k - j), val(o.idx, j))));
                                }
                                const auto ps = conv_sum(std::move(terms));
                                const auto ps2 = def(ps + " + " + ps);
                                sq = def(ps2 + " + " + sq);
                            }
                            tmp.push_back(std::move(sq));
                        } else if (k == 0u) {
                            tmp.push_back(def(mul(numpar(o), numpar(o))));
                        } else {
                            tmp.emplace_back("0.0");
                        }
                    }
                    out = pairwise_sum(std::move(tmp));
                }
                break;
            }
            case func_kind::pow: {
                // Reference: src/math/pow.cpp:395-550.
                if (a[1].type != operand::kind::num) {
                    throw std::invalid_argument("An invalid argument type was encountered while trying to build "
                                                "the Taylor derivative of a pow()");
                }
                const auto ex = a[1].value;
                if (!is_var(a[0])) {
                    out = (k == 0u) ? pow_eval(numpar(a[0]), ex) : "0.0";
                    break;
                }
                const auto b = a[0].idx;
                if (k == 0u) {
                    out = pow_eval(val(b, 0), ex);
                } else if (ex == .5) {
                    // sqrt() special case.
                    const auto dv = def(val(u, 0) + " + " + val(u, 0));
                    std::string fac = val(b, k);
                    std::vector<std::string> terms;
                    const auto jmax = (k % 2u == 1u) ? (k - 1u) / 2u : (k - 2u) / 2u;
                    for (std::uint32_t j = 1; j <= jmax; ++j) {
                        terms.push_back(def(mul(val(u, k - j), val(u, j))));
                    }
                    if (k % 2u == 0u) {
                        const auto &hv = val(u, k / 2u);
                        const auto sq = def(mul(hv, hv));
                        fac = def(fac + " - " + sq);
                    }
                    if (!terms.empty()) {
                        const auto ps = conv_sum(std::move(terms));
                        const auto ps2 = def(ps + " + " + ps);
                        fac = def(fac + " - " + ps2);
                    }
                    out = def(fac + " / " + dv);
                } else if (ex == 2.) {
                    // square() special case.
                    std::vector<std::string> terms;
                    if (k % 2u == 1u) {
                        for (std::uint32_t j = 0; j <= (k - 1u) / 2u; ++j) {
                            terms.push_back(def(mul(val(b, k - j), val(b, j))));
                        }
                        const auto ps = conv_sum(std::move(terms));
                        out = def(ps + " + " + ps);
                    } else {
                        const auto &hv = val(b, k / 2u);
                        const auto sq = def(mul(hv, hv));
                        for (std::uint32_t j = 0; j <= (k - 2u) / 2u; ++j) {
                            terms.push_back(def(mul(val(b, k - j), val(b, j))));
                        }
                        const auto ps = conv_sum(std::move(terms));
                        const auto ps2 = def(ps + " + " + ps);
                        out = def(ps2 + " + " + sq);
                    }
                } else {
                    std::vector<std::string> terms;
                    for (std::uint32_t j = 0; j < k; ++j) {
                        // Scalar factor folded in double like the reference's IR constants:
                        // order * alpha - j * (alpha + 1).
                        const double sf = static_cast<double>(k) * ex - static_cast<double>(j) * (ex + 1.);
                        const auto pr = def(mul(val(b, k - j), val(u, j)));
                        terms.push_back(def(mul(fp_literal(sf), pr)));
                    }
                    const auto acc = conv_sum(std::move(terms));
                    out = pow_quotient(u, b, acc, k);
                }
                break;
            }
            case func_kind::sin:
            case func_kind::cos: {
                // Reference: src/math/sin.cpp:152-192, src/math/cos.cpp:152-185.
                const bool is_sin = n.kind == func_kind::sin;
                if (!is_var(a[0])) {
                    out = (k == 0u) ? def(std::string(is_sin ? "hy_sin(" : "hy_cos(") + numpar(a[0]) + ")") : "0.0";
                    break;
                }
                const auto b = a[0].idx;
                if (k == 0u) {
                    out = def(std::string(is_sin ? "hy_sin(" : "hy_cos(") + val(b, 0) + ")");
                } else {
                    if (n.deps.size() != 1u) {
                        throw std::invalid_argument("A hidden dependency vector of size 1 is expected in order to "
                                                    "compute the Taylor derivative of the sine/cosine");
                    }
                    const auto d = n.deps[0];
                    std::vector<std::string> terms;
                    for (std::uint32_t j = 1; j <= k; ++j) {
                        const auto pr = def(mul(val(d, k - j), val(b, j)));
                        terms.push_back(def(mul(fp_literal(static_cast<double>(j)), pr)));
                    }
                    const auto acc = conv_sum(std::move(terms));
                    // NOTE: division by the (constant) order via the exact FMA sequence of div_const() (bit-identical
                    // to the IEEE quotient); x / (-k) == -(x / k) exactly.
                    const auto q = div_const(acc, k);
                    out = is_sin ? q : def("-" + q);
                }
                break;
            }
            case func_kind::exp: {
                // Reference: src/math/exp.cpp:84-120.
                if (!is_var(a[0])) {
                    out = (k == 0u) ? def("exp(" + numpar(a[0]) + ")") : "0.0";
                    break;
                }
                const auto b = a[0].idx;
                if (k == 0u) {
                    out = def("exp(" + val(b, 0) + ")");
                } else {
                    std::vector<std::string> terms;
                    for (std::uint32_t j = 1; j <= k; ++j) {
                        const auto pr = def(mul(val(u, k - j), val(b, j)));
                        terms.push_back(def(mul(fp_literal(static_cast<double>(j)), pr)));
                    }
                    const auto acc = conv_sum(std::move(terms));
                    out = div_const(acc, k);
                }
                break;
            }
            case func_kind::log: {
                // Reference: src/math/log.cpp (taylor_diff_log_impl).
                if (!is_var(a[0])) {
                    out = (k == 0u) ? def("log(" + numpar(a[0]) + ")") : "0.0";
                    break;
                }
                const auto b = a[0].idx;
                if (k == 0u) {
                    out = def("log(" + val(b, 0) + ")");
                } else {
                    const auto kf = fp_literal(static_cast<double>(k));
                    const auto nb0 = def(mul(kf, val(b, 0)));
                    auto ret = def(mul(kf, val(b, k)));
                    if (k > 1u) {
                        std::vector<std::string> terms;
                        for (std::uint32_t j = 1; j < k; ++j) {
                            const auto pr = def(mul(val(b, k - j), val(u, j)));
                            terms.push_back(def(mul(fp_literal(static_cast<double>(j)), pr)));
                        }
                        const auto acc = conv_sum(std::move(terms));
                        ret = def(ret + " - " + acc);
                    }
                    out = def(ret + " / " + nb0);
                }
                break;
            }
            case func_kind::tan:
            case func_kind::tanh:
            case func_kind::sinh:
            case func_kind::cosh:
            case func_kind::erf:
            case func_kind::sigmoid: {
                // "Forward" rules: k a^[k] = sum_{j=1..k} j X^[k-j] b^[j], with
                //   tan:  a^[k] = b^[k] + S/k, X = a^2 (hidden dep)            src/math/tan.cpp:105-131
                //   tanh: a^[k] = b^[k] - S/k, X = a^2                          src/math/tanh.cpp
                //   sinh / cosh: a^[k] = S/k, X = the partner function           src/math/sinh.cpp, cosh.cpp
                //   erf:  a^[k] = 2/sqrt(pi) S/k, X = exp(-b^2)                  src/math/erf.cpp
                //   sigmoid: a^[k] = S/k, X = a - a^2                            src/math/sigmoid.cpp:150-172
                const auto fname = std::string(func_kind_name(n.kind));
                const auto order0 = [&](const std::string &x) {
                    return n.kind == func_kind::sigmoid ? def("1.0 / (1.0 + exp(-(" + x + ")))") : def("hy_" + fname + "(" + x + ")");
                };
                if (!is_var(a[0])) {
                    out = (k == 0u) ? order0(numpar(a[0])) : "0.0";
                    break;
                }
                const auto b = a[0].idx;
                if (k == 0u) {
                    out = order0(val(b, 0));
                    break;
                }
                if (n.deps.size() != 1u) {
                    throw std::invalid_argument("A hidden dependency vector of size 1 is expected in order to compute "
                                                "the Taylor derivative of " + fname);
                }
                const auto d = n.deps[0];
                std::vector<std::string> terms;
                for (std::uint32_t j = 1; j <= k; ++j) {
                    std::string x = val(d, k - j);
                    if (n.kind == func_kind::sigmoid) {
                        x = def(val(u, k - j) + " - " + x);
                    }
                    const auto pr = def(mul(x, val(b, j)));
                    terms.push_back(def(mul(fp_literal(static_cast<double>(j)), pr)));
                }
                const auto acc = div_const(conv_sum(std::move(terms)), k);
                if (n.kind == func_kind::tan) {
                    out = def(val(b, k) + " + " + acc);
                } else if (n.kind == func_kind::tanh) {
                    out = def(val(b, k) + " - " + acc);
                } else if (n.kind == func_kind::erf) {
                    // 2 / sqrt(pi)
                    out = def(mul(fp_literal(1.1283791670955125738961589031215451716881012586580), acc));
                } else {
                    out = acc;
                }
                break;
            }
            case func_kind::asin:
            case func_kind::acos:
            case func_kind::atan:
            case func_kind::asinh:
            case func_kind::acosh:
            case func_kind::atanh: {
                // "Inverse" rules: a^[k] = (k b^[k] -+ sum_{j=1..k-1} j c^[k-j] a^[j]) / (k D), with c the hidden dep and
                //   asin:  D = c0 = sqrt(1 - b0^2), minus            src/math/asin.cpp:140-178
                //   acos:  D = -c0, plus                              src/math/acos.cpp:140-178
                //   atan:  D = 1 + c0 (c = b^2), minus                src/math/atan.cpp:125-164
                //   atanh: D = 1 - c0 (c = b^2), plus                 src/math/atanh.cpp:125-164
                //   asinh: D = c0 = sqrt(1 + b0^2), minus             src/math/asinh.cpp:135-170
                //   acosh: D = c0 = sqrt(b0^2 - 1), minus             src/math/acosh.cpp
                const auto fname = std::string(func_kind_name(n.kind));
                if (!is_var(a[0])) {
                    out = (k == 0u) ? def("hy_" + fname + "(" + numpar(a[0]) + ")") : "0.0";
                    break;
                }
                const auto b = a[0].idx;
                if (k == 0u) {
                    out = def("hy_" + fname + "(" + val(b, 0) + ")");
                    break;
                }
                if (n.deps.size() != 1u) {
                    throw std::invalid_argument("A hidden dependency vector of size 1 is expected in order to compute "
                                                "the Taylor derivative of " + fname);
                }
                const auto d = n.deps[0];
                std::string D;
                switch (n.kind) {
                    case func_kind::acos:
                        D = def("-" + val(d, 0));
                        break;
                    case func_kind::atan:
                        D = def(val(d, 0) + " + 1.0");
                        break;
                    case func_kind::atanh:
                        D = def("1.0 - " + val(d, 0));
                        break;
                    default:
                        D = val(d, 0);
                        break;
                }
                if (k == 1u) {
                    out = def(val(b, 1) + " / " + D);
                    break;
                }
                const bool plus = (n.kind == func_kind::acos || n.kind == func_kind::atanh);
                const auto kf = fp_literal(static_cast<double>(k));
                auto ret = def(mul(kf, val(b, k)));
                std::vector<std::string> terms;
                for (std::uint32_t j = 1; j < k; ++j) {
                    const auto pr = def(mul(val(d, k - j), val(u, j)));
                    terms.push_back(def(mul(fp_literal(static_cast<double>(j)), pr)));
                }
                ret = def(ret + (plus ? " + " : " - ") + conv_sum(std::move(terms)));
                out = def(ret + " / " + def(mul(kf, D)));
                break;
            }
            case func_kind::relu:
            case func_kind::relup: {
                // relu(b)^[k] = b^[0] > 0 ? b^[k] : slope * b^[k] (all orders), relup(b)^[0] = b^[0] > 0 ? 1 : slope and 0
                // beyond (src/math/relu.cpp:144-178, :392-424).
                const auto slope = a.at(1).value;
                const auto pick = [&](const std::string &c, const std::string &x) {
                    if (n.kind == func_kind::relup) {
                        return def("(" + c + " > 0.0) ? 1.0 : " + fp_literal(slope));
                    }
                    return slope == 0. ? def("(" + c + " > 0.0) ? " + x + " : 0.0")
                                       : def("(" + c + " > 0.0) ? " + x + " : " + def(mul(fp_literal(slope), x)));
                };
                if (!is_var(a[0])) {
                    out = (k == 0u) ? pick(numpar(a[0]), numpar(a[0])) : "0.0";
                } else if (n.kind == func_kind::relup) {
                    out = (k == 0u) ? pick(val(a[0].idx, 0), "") : "0.0";
                } else {
                    out = pick(val(a[0].idx, 0), val(a[0].idx, k));
                }
                break;
            }
            case func_kind::select: {
                // select(c, t, f)^[k] = c^[0] != 0 ? t^[k] : f^[k] (src/math/select.cpp:88-131).
                const auto arg_k = [&](const operand &o) {
                    return is_var(o) ? val(o.idx, k) : (k == 0u ? numpar(o) : std::string("0.0"));
                };
                const auto c0 = is_var(a[0]) ? val(a[0].idx, 0) : numpar(a[0]);
                out = def("(" + c0 + " != 0.0) ? " + arg_k(a.at(1)) + " : " + arg_k(a.at(2)));
                break;
            }
            case func_kind::logical_and:
            case func_kind::logical_or:
            case func_kind::rel_eq:
            case func_kind::rel_neq:
            case func_kind::rel_lt:
            case func_kind::rel_gt:
            case func_kind::rel_lte:
            case func_kind::rel_gte: {
                // Truth values 1 / 0 at order 0, zero beyond (src/math/relational.cpp:197-238, logical.cpp:93-127, :266-300).
                if (k != 0u) {
                    out = "0.0";
                    break;
                }
                const auto arg0 = [&](const operand &o) { return is_var(o) ? val(o.idx, 0) : numpar(o); };
                std::string e;
                if (n.kind == func_kind::logical_and || n.kind == func_kind::logical_or) {
                    for (std::size_t i = 0; i < a.size(); ++i) {
                        e += (i == 0u ? "" : (n.kind == func_kind::logical_and ? " & " : " | "));
                        e += "(" + arg0(a[i]) + " != 0.0)";
                    }
                } else {
                    // NOTE: "neq" is an ordered comparison in the reference (false if an operand is nan).
                    const auto x = arg0(a.at(0)), y = arg0(a.at(1));
                    switch (n.kind) {
                        case func_kind::rel_eq: e = x + " == " + y; break;
                        case func_kind::rel_neq: e = "(" + x + " < " + y + ") | (" + x + " > " + y + ")"; break;
                        case func_kind::rel_lt: e = x + " < " + y; break;
                        case func_kind::rel_gt: e = x + " > " + y; break;
                        case func_kind::rel_lte: e = x + " <= " + y; break;
                        default: e = x + " >= " + y; break;
                    }
                }
                out = def("(" + e + ") ? 1.0 : 0.0");
                break;
            }
            case func_kind::atan2: {
                // a = atan2(b, c), d = b^2 + c^2 (hidden dependency). Reference: src/math/atan2.cpp:113-330:
                //   a^[k] = (k (c^[0] b^[k] - b^[0] c^[k]) + sum_{j=1..k-1} j (c^[k-j] b^[j] - b^[k-j] c^[j] - d^[k-j] a^[j]))
                //           / (k d^[0]),
                // with the terms of a constant argument dropped.
                const bool vy = is_var(a[0]), vx = is_var(a[1]);
                const auto arg0 = [&](const operand &o) { return is_var(o) ? val(o.idx, 0) : numpar(o); };
                if (k == 0u) {
                    out = def("hy_atan2(" + arg0(a[0]) + ", " + arg0(a[1]) + ")");
                    break;
                }
                if (!vy && !vx) {
                    out = "0.0";
                    break;
                }
                const auto d = n.deps.at(0);
                const auto kf = fp_literal(static_cast<double>(k));
                const auto divisor = def(mul(kf, val(d, 0)));
                std::string dividend;
                if (vy && vx) {
                    const auto b = a[0].idx, c = a[1].idx;
                    const auto t1 = def(mul(val(c, 0), val(b, k)));
                    const auto t2 = def(mul(val(b, 0), val(c, k)));
                    dividend = def(mul(kf, def(t1 + " - " + t2)));
                } else if (vy) {
                    dividend = def(mul(kf, def(mul(numpar(a[1]), val(a[0].idx, k)))));
                } else {
                    dividend = def(mul(fp_literal(-static_cast<double>(k)), def(mul(numpar(a[0]), val(a[1].idx, k)))));
                }
                if (k > 1u) {
                    std::vector<std::string> terms;
                    for (std::uint32_t j = 1; j < k; ++j) {
                        const auto t3 = def(mul(val(d, k - j), val(u, j)));
                        if (vy && vx) {
                            const auto b = a[0].idx, c = a[1].idx;
                            const auto t1 = def(mul(val(c, k - j), val(b, j)));
                            const auto t2 = def(mul(val(b, k - j), val(c, j)));
                            const auto t = def(def(t1 + " - " + t2) + " - " + t3);
                            terms.push_back(def(mul(fp_literal(static_cast<double>(j)), t)));
                        } else {
                            terms.push_back(def(mul(fp_literal(-static_cast<double>(j)), t3)));
                        }
                    }
                    dividend = def(dividend + " + " + conv_sum(std::move(terms)));
                }
                out = def(dividend + " / " + divisor);
                break;
            }
            case func_kind::custom: {
                // A function defined through the registry of node rules (node_rule.hpp): order 0 from the values of the
                // arguments; order k from jets - here local arrays of the SSA values (the rule is force-inlined with a
                // constant k: its loops unroll and the arrays dissolve into registers). Arguments: orders 0 .. k (a
                // number / parameter: order 0 only), the node itself and its hidden dependencies: orders 0 .. k - 1.
                const auto &rule = get_node_rule(n.rule);
                const auto uid = std::to_string(counter++);
                if (k == 0u) {
                    // (Out of line, arguments by value: node_rules_device_source().)
                    std::string call = "hy_rule_" + rule.name + "_value(";
                    for (std::size_t i2 = 0; i2 < a.size(); ++i2) {
                        call += (i2 == 0u ? "" : ", ") + (is_var(a[i2]) ? val(a[i2].idx, 0) : numpar(a[i2]));
                    }
                    out = def(call + ")");
                    break;
                }
                const auto arr = [&](const std::string &name, const std::function<std::string(std::uint32_t)> &coeff,
                                     std::uint32_t n_valid) {
                    os << "const double " << name << "[] = {";
                    for (std::uint32_t j = 0; j < n_valid; ++j) {
                        os << (j == 0u ? "" : ", ") << coeff(j);
                    }
                    os << "};\n";
                    return "{" + name + ", 1u, " + std::to_string(n_valid) + "u}";
                };
                std::string xj, hj;
                for (std::size_t i2 = 0; i2 < a.size(); ++i2) {
                    const auto &o = a[i2];
                    const auto nm = "xa" + uid + "_" + std::to_string(i2);
                    xj += (i2 == 0u ? "" : ", ")
                          + (is_var(o) ? arr(nm, [&](std::uint32_t j) { return val(o.idx, j); }, k + 1u)
                                       : arr(nm, [&](std::uint32_t) { return numpar(o); }, 1u));
                }
                for (std::size_t i2 = 0; i2 < n.deps.size(); ++i2) {
                    const auto d = n.deps[i2];
                    hj += (i2 == 0u ? "" : ", ") + arr("ha" + uid + "_" + std::to_string(i2), [&](std::uint32_t j) { return val(d, j); }, k);
                }
                const auto self = arr("sa" + uid, [&](std::uint32_t j) { return val(u, j); }, k);
                os << "const hy_jet xj" << uid << "[] = {" << (xj.empty() ? "{nullptr, 1u, 0u}" : xj) << "};\n";
                os << "const hy_jet hj" << uid << "[] = {" << (hj.empty() ? "{nullptr, 1u, 0u}" : hj) << "};\n";
                out = def("hy_rule_" + rule.name + "_orderk(" + std::to_string(k) + "u, hy_jet" + self + ", xj" + uid + ", hj" + uid
                          + ")");
                break;
            }
            case func_kind::kepE: {
                // a = E(e, M) with E - e sin E = M; hidden dependencies c = e cos(a), d = sin(a), in this order.
                // Reference: src/math/kepE.cpp:140-355:
                //   a^[k] = (k (e^[k] d^[0] + M^[k]) + sum_{j=1..k-1} j (c^[k-j] a^[j] + d^[k-j] e^[j])) / (k (1 - c^[0])),
                // with the terms of a constant argument dropped.
                const bool ve = is_var(a[0]), vm = is_var(a[1]);
                const auto arg0 = [&](const operand &o) { return is_var(o) ? val(o.idx, 0) : numpar(o); };
                if (k == 0u) {
                    out = def("hy_kepE(" + arg0(a[0]) + ", " + arg0(a[1]) + ")");
                    break;
                }
                if (!ve && !vm) {
                    out = "0.0";
                    break;
                }
                const auto c = n.deps.at(0), d = n.deps.at(1);
                const auto kf = fp_literal(static_cast<double>(k));
                const auto divisor = def(mul(kf, def("1.0 - " + val(c, 0))));
                std::string dividend;
                if (ve && vm) {
                    const auto t = def(def(mul(val(a[0].idx, k), val(d, 0))) + " + " + val(a[1].idx, k));
                    dividend = def(mul(kf, t));
                } else if (ve) {
                    dividend = def(mul(kf, def(mul(val(a[0].idx, k), val(d, 0)))));
                } else {
                    dividend = def(mul(kf, val(a[1].idx, k)));
                }
                if (k > 1u) {
                    std::vector<std::string> terms;
                    for (std::uint32_t j = 1; j < k; ++j) {
                        const auto jf = fp_literal(static_cast<double>(j));
                        const auto ca = def(mul(val(c, k - j), val(u, j)));
                        if (ve) {
                            const auto de = def(mul(val(d, k - j), val(a[0].idx, j)));
                            terms.push_back(def(mul(jf, def(ca + " + " + de))));
                        } else {
                            terms.push_back(def(mul(jf, ca)));
                        }
                    }
                    dividend = def(dividend + " + " + conv_sum(std::move(terms)));
                }
                out = def(dividend + " / " + divisor);
                break;
            }
        }
    }

    // Order-k coefficient of state variable i (reference: taylor_compute_sv_diff(),
    // src/taylor_02.cpp:245-287: true division by the order).
    void sv(std::uint32_t i, std::uint32_t k)
    {
        assert(k > 0u);
        const auto &d = p.sv_defs[i];
        auto &out = val(i, k);
        if (d.type == operand::kind::uvar) {
            // NOTE: x^[k] = rhs^[k-1] / k is a true division in the reference (src/taylor_02.cpp:266-268); div_const()
            // produces the same correctly-rounded quotient with 3 multiply-type operations instead of the ~10
            // instructions of a division sequence (two-body kernel: 112 divisions per step, 23 % of the instructions).
            out = div_const(val(d.idx, k - 1u), k);
        } else {
            out = (k == 1u) ? numpar(d) : "0.0";
        }
    }

    // ---------------------------------------------------------------------------------------------
    // Software-pipelined ("split") evaluation, used by the cluster generator.
    //
    // The order-k coefficient of a convolution-type node is split into a *history part* (all terms
    // that involve only coefficients of order < k, which are available one full phase earlier) and a
    // short *finish* (the 1-2 terms containing order-k operands). The history part is accumulated
    // with an FMA chain (like the running sums of the reference's compact mode,
    // src/math/prod.cpp:686-698) and can be scheduled under the latency of the LDS exchange.
    // ---------------------------------------------------------------------------------------------
    std::map<std::pair<std::uint32_t, std::uint32_t>, std::string> partials;

    // acc += a * b as a chain (contracted into an FMA by -ffp-contract=fast).
    std::string chain(const std::string &acc, const std::string &a, const std::string &b)
    {
        if (acc.empty()) {
            return def(mul(a, b));
        }
        return def(a + " * " + b + " + " + acc);
    }

    // Is the split form implemented for node i?
    bool can_split(std::uint32_t i) const
    {
        const auto &n = p.nodes[i];
        const auto &a = n.args;
        switch (n.kind) {
            case func_kind::prod:
                return a.size() == 2u && is_var(a[0]) && is_var(a[1]) && cu[a[0].idx] == 0 && cu[a[1].idx] == 0;
            case func_kind::sum_sq:
                for (const auto &o : a) {
                    if (!is_var(o)) {
                        return false;
                    }
                }
                return true;
            case func_kind::pow:
                return is_var(a[0]) && a[1].type == operand::kind::num && a[1].value != .5 && a[1].value != 2.;
            default:
                return false;
        }
    }

    // One term of a history chain: scale * (a * b) (no scaling if scale is empty).
    struct chain_term {
        std::string a, b, scale;
    };

    // Terms of the history part of the order-k coefficient of node i (k >= 2): orders 1..k-1 of the
    // operands (and of the node itself, for pow). sq receives the "middle squares" of sum_sq at even orders.
    void partial_terms(std::uint32_t i, std::uint32_t k, std::vector<chain_term> &main, std::vector<chain_term> &sq)
    {
        const auto &n = p.nodes[i];
        const auto u = p.n_eq + i;
        const auto &a = n.args;
        switch (n.kind) {
            case func_kind::prod:
                for (std::uint32_t j = 1; j < k; ++j) {
                    main.push_back({val(a[0].idx, k - j), val(a[1].idx, j), {}});
                }
                break;
            case func_kind::sum_sq: {
                const auto jmax = (k % 2u == 1u) ? (k - 1u) / 2u : (k - 2u) / 2u;
                for (std::uint32_t j = 1; j <= jmax; ++j) {
                    for (const auto &o : a) {
                        main.push_back({val(o.idx, k - j), val(o.idx, j), {}});
                    }
                }
                if (k % 2u == 0u) {
                    for (const auto &o : a) {
                        sq.push_back({val(o.idx, k / 2u), val(o.idx, k / 2u), {}});
                    }
                }
                break;
            }
            case func_kind::pow: {
                const auto ex = a[1].value;
                for (std::uint32_t j = 1; j < k; ++j) {
                    const double sf = static_cast<double>(k) * ex - static_cast<double>(j) * (ex + 1.);
                    main.push_back({val(a[0].idx, k - j), val(u, j), fp_literal(sf)});
                }
                break;
            }
            default:
                break;
        }
    }

    // Emit the history parts of order k for a set of nodes, interleaving the FMA chains of the different
    // nodes term by term (independent chains back to back hide the FP64 FMA latency at one wave per SIMD).
    // NOTE: the interleaved chains can be emitted in several parts (part = 0 .. n_parts - 1), so
    // that the caller can spread the history work over different latency windows. The accumulators
    // are carried from one part to the next in pending_chains.
    struct chain_state {
        std::uint32_t node;
        bool is_sq;
        std::vector<chain_term> terms;
        std::string acc;
    };
    std::map<std::uint32_t, std::vector<chain_state>> pending_chains;

    void emit_partials(const std::vector<std::uint32_t> &node_ids, std::uint32_t k, std::uint32_t part = 0,
                       std::uint32_t n_parts = 1)
    {
        if (k < 2u) {
            return;
        }
        if (part == 0u) {
            std::vector<chain_state> chains;
            for (const auto i : node_ids) {
                if (!can_split(i)) {
                    continue;
                }
                std::vector<chain_term> main, sq;
                partial_terms(i, k, main, sq);
                chains.push_back({i, false, std::move(main), {}});
                if (!sq.empty()) {
                    chains.push_back({i, true, std::move(sq), {}});
                }
            }
            pending_chains[k] = std::move(chains);
        }
        auto &chains = pending_chains[k];
        std::size_t max_len = 0;
        for (const auto &c : chains) {
            max_len = std::max(max_len, c.terms.size());
        }
        const auto t_begin = max_len * part / n_parts;
        const auto t_end = max_len * (part + 1u) / n_parts;
        for (std::size_t t = t_begin; t < t_end; ++t) {
            for (auto &c : chains) {
                if (t >= c.terms.size()) {
                    continue;
                }
                const auto &tm = c.terms[t];
                if (tm.scale.empty()) {
                    c.acc = chain(c.acc, tm.a, tm.b);
                } else {
                    const auto pr = def(mul(tm.a, tm.b));
                    c.acc = chain(c.acc, tm.scale, pr);
                }
            }
        }
        if (part + 1u == n_parts) {
            for (auto &c : chains) {
                partials[{c.node, c.is_sq ? (k + 0x10000u) : k}] = c.acc;
            }
            pending_chains.erase(k);
        }
    }

    // ---- Selective emission of the history chains (explicit overlap with LDS latency windows) ----
    // The terms of the order-K history chain of a node are classified as
    //   late  - they involve an order-(K-1) coefficient (available only once order K-1 is complete),
    //   early - both coefficients have order <= K-2 (available one full order earlier);
    // the early terms are further split in two halves (early_a / early_b). The caller emits the three
    // selections at three different points of the schedule (see hip_emit_cluster2.cpp); the accumulators
    // are carried in pending_sel and published to `partials` when the last selection has been emitted.
    // NOTE: the order of accumulation within a chain is early terms first, then late terms.
    enum class part_sel { early_a = 0, early_b = 1, late = 2 };

    struct idx_term {
        std::uint32_t ua, ka, ub, kb;
        std::string scale;
    };
    struct sel_chain {
        std::uint32_t node;
        bool is_sq;
        std::vector<idx_term> early, late;
        std::string acc;
    };
    struct sel_state {
        std::vector<sel_chain> chains;
        unsigned done_mask = 0;
    };
    std::map<std::uint32_t, sel_state> pending_sel;
    // Percentage of the early terms emitted with the early_a selection.
    std::size_t early_a_pct = 50;

    void emit_partials_sel(const std::vector<std::uint32_t> &node_ids, std::uint32_t K, part_sel sel)
    {
        if (K < 2u || K >= order) {
            return;
        }
        auto it = pending_sel.find(K);
        if (it == pending_sel.end()) {
            sel_state st;
            for (const auto i : node_ids) {
                if (!can_split(i)) {
                    continue;
                }
                const auto &n = p.nodes[i];
                const auto u = p.n_eq + i;
                const auto &a = n.args;
                sel_chain mainc{i, false, {}, {}, {}}, sqc{i, true, {}, {}, {}};
                const auto add = [&](sel_chain &c, idx_term t) {
                    ((t.ka == K - 1u || t.kb == K - 1u) ? c.late : c.early).push_back(std::move(t));
                };
                switch (n.kind) {
                    case func_kind::prod:
                        for (std::uint32_t j = 1; j < K; ++j) {
                            add(mainc, {a[0].idx, K - j, a[1].idx, j, {}});
                        }
                        break;
                    case func_kind::sum_sq: {
                        const auto jmax = (K % 2u == 1u) ? (K - 1u) / 2u : (K - 2u) / 2u;
                        for (std::uint32_t j = 1; j <= jmax; ++j) {
                            for (const auto &o : a) {
                                add(mainc, {o.idx, K - j, o.idx, j, {}});
                            }
                        }
                        if (K % 2u == 0u) {
                            for (const auto &o : a) {
                                add(sqc, {o.idx, K / 2u, o.idx, K / 2u, {}});
                            }
                        }
                        break;
                    }
                    case func_kind::pow: {
                        const auto ex = a[1].value;
                        for (std::uint32_t j = 1; j < K; ++j) {
                            const double sf = static_cast<double>(K) * ex - static_cast<double>(j) * (ex + 1.);
                            add(mainc, {a[0].idx, K - j, u, j, fp_literal(sf)});
                        }
                        break;
                    }
                    default:
                        break;
                }
                const bool has_sq = !sqc.early.empty() || !sqc.late.empty();
                st.chains.push_back(std::move(mainc));
                if (has_sq) {
                    st.chains.push_back(std::move(sqc));
                }
            }
            it = pending_sel.emplace(K, std::move(st)).first;
        }
        auto &st = it->second;
        // Interleave the chains term by term.
        const auto range_of = [&](const sel_chain &c) -> std::pair<std::size_t, std::size_t> {
            switch (sel) {
                case part_sel::early_a:
                    return {0u, c.early.size() * early_a_pct / 100u};
                case part_sel::early_b:
                    return {c.early.size() * early_a_pct / 100u, c.early.size()};
                default:
                    return {0u, c.late.size()};
            }
        };
        std::size_t max_len = 0;
        for (const auto &c : st.chains) {
            const auto [b, e] = range_of(c);
            max_len = std::max(max_len, e - b);
        }
        for (std::size_t t = 0; t < max_len; ++t) {
            for (auto &c : st.chains) {
                const auto [b, e] = range_of(c);
                if (b + t >= e) {
                    continue;
                }
                const auto &tm = (sel == part_sel::late) ? c.late[b + t] : c.early[b + t];
                const auto &va = val(tm.ua, tm.ka);
                const auto &vb = val(tm.ub, tm.kb);
                assert(!va.empty() && !vb.empty());
                if (tm.scale.empty()) {
                    c.acc = chain(c.acc, va, vb);
                } else {
                    const auto pr = def(mul(va, vb));
                    c.acc = chain(c.acc, tm.scale, pr);
                }
            }
        }
        st.done_mask |= 1u << static_cast<unsigned>(sel);
        if (st.done_mask == 7u) {
            for (auto &c : st.chains) {
                partials[{c.node, c.is_sq ? (K + 0x10000u) : K}] = c.acc;
            }
            pending_sel.erase(it);
        }
    }

    // History part of a single node (non-interleaved form).
    void node_partial(std::uint32_t i, std::uint32_t k)
    {
        emit_partials({i}, k);
    }

    // Order-k coefficient of node i using the history part emitted earlier (falls back to node()).
    void node_finish(std::uint32_t i, std::uint32_t k)
    {
        const auto it = partials.find({i, k});
        if (it == partials.end()) {
            node(i, k);
            return;
        }
        const auto &n = p.nodes[i];
        const auto u = p.n_eq + i;
        const auto &a = n.args;
        auto acc = it->second;
        auto &out = val(u, k);
        switch (n.kind) {
            case func_kind::prod:
                acc = chain(acc, val(a[0].idx, k), val(a[1].idx, 0));
                out = chain(acc, val(a[0].idx, 0), val(a[1].idx, k));
                break;
            case func_kind::sum_sq: {
                for (const auto &o : a) {
                    acc = chain(acc, val(o.idx, k), val(o.idx, 0));
                }
                const auto dbl = def(acc + " + " + acc);
                if (k % 2u == 0u) {
                    out = def(dbl + " + " + partials.at({i, k + 0x10000u}));
                } else {
                    out = dbl;
                }
                break;
            }
            case func_kind::pow: {
                const auto ex = a[1].value;
                const double sf = static_cast<double>(k) * ex;
                const auto pr = def(mul(val(a[0].idx, k), val(u, 0)));
                acc = chain(acc, fp_literal(sf), pr);
                out = pow_quotient(u, a[0].idx, acc, k);
                break;
            }
            default:
                break;
        }
    }

    // Quotient acc / (k b_0) of the pow recurrence (src/math/pow.cpp:546-549) without a division sequence per order
    // (~13 instructions on gfx950): r = RN(1 / b_0) once per step and node, then q0 = acc * r_k with r_k = r * RN(1 / k),
    // the exact residual rem = acc - (k b_0) q0 (FMA) and q = q0 + rem * r_k (Markstein: the correctly-rounded quotient
    // unless r_k is off by more than an ulp in a halfway case). Opt-in of the emitters whose step body is ONE scope (the
    // reciprocal is defined at the first use and read at every later order): enable_pow_rcp(); kw::exact_division
    // keeps the plain division.
    bool pow_rcp = false;
    void enable_pow_rcp(bool on = true)
    {
        pow_rcp = on;
    }
    std::map<std::uint32_t, std::string> pow_r0;
    std::string pow_quotient(std::uint32_t u, std::uint32_t b, const std::string &acc, std::uint32_t k)
    {
        const auto dv = def(mul(fp_literal(static_cast<double>(k)), val(b, 0)));
        if (!pow_rcp) {
            return def(acc + " / " + dv);
        }
        auto &r = pow_r0[u];
        if (r.empty()) {
            r = def("1.0 / " + val(b, 0));
        }
        const auto rk = (k == 1u) ? r : def(mul(r, fp_literal(1. / static_cast<double>(k))));
        const auto q0 = def(mul(acc, rk));
        const auto rem = def("__builtin_fma(-" + dv + ", " + q0 + ", " + acc + ")");
        return def("__builtin_fma(" + rem + ", " + rk + ", " + q0 + ")");
    }

    // x / d for a small integer constant d > 0 without a division: q = RN(x * r), r = RN(1 / d);
    // rem = x - q * d (exact, FMA); result = RN(q + rem * r). By Markstein's theorem the result is the
    // correctly-rounded quotient, i.e. bit-identical to IEEE x / d (outside of the subnormal range).
    // (recip_div: the lane-pair kernel's choice - one multiplication by RN(1 / d), within 1 ulp.)
    bool recip_div = false;
    std::string div_const(const std::string &x, std::uint32_t d)
    {
        if (d == 1u) {
            return x;
        }
        if (fold_zeros && is_zero_lit(x)) {
            return "0.0";
        }
        if ((d & (d - 1u)) == 0u || recip_div) {
            // Power of two: the multiplication by the reciprocal is exact.
            return def(mul(x, fp_literal(1. / static_cast<double>(d))));
        }
        const auto r = fp_literal(1. / static_cast<double>(d));
        const auto q = def(mul(x, r));
        const auto rem = def("__builtin_fma(" + fp_literal(-static_cast<double>(d)) + ", " + q + ", " + x + ")");
        return def("__builtin_fma(" + rem + ", " + r + ", " + q + ")");
    }
};

} // namespace heyoka_amd::emit_detail
