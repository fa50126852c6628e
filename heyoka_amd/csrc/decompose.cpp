// Taylor decomposition implementation. See decompose.hpp for the reference citations.
#include "decompose.hpp"

#include <algorithm>
#include <cassert>
#include <charconv>
#include <cmath>
#include <deque>
#include <limits>
#include <optional>
#include <set>
#include <unordered_map>
#include <unordered_set>

namespace heyoka_amd
{

std::uint32_t uname_to_index(const std::string &s)
{
    assert(s.size() > 2u && s[0] == 'u' && s[1] == '_');
    std::uint32_t value = 0;
    const auto ret = std::from_chars(s.data() + 2, s.data() + s.size(), value);
    if (ret.ec != std::errc{} || ret.ptr != s.data() + s.size()) {
        throw std::invalid_argument("Cannot extract a u variable index from the string '" + s + "'");
    }
    return value;
}

namespace
{

std::string uname(std::size_t i)
{
    return "u_" + std::to_string(i);
}

bool is_negative_one(const expression &e)
{
    return e.is_number() && e.num() == -1;
}

bool is_kind(const expression &e, func_kind k)
{
    return e.is_func() && e.fn().kind() == k;
}

// Apply a branch transformation to every expression of a vector with a common cache.
std::vector<expression> transform_all(const std::vector<expression> &v_ex,
                                      const std::function<expression(const expression &)> &tfunc)
{
    ptr_ex_map cache;
    std::vector<expression> ret;
    ret.reserve(v_ex.size());
    for (const auto &e : v_ex) {
        ret.push_back(traverse_transform_nodes(cache, e, {}, tfunc));
    }
    return ret;
}

// x**y -> exp(y*log(x)) if y is not a number (reference: src/taylor_01.cpp:806-840).
std::vector<expression> pow_to_explog(const std::vector<expression> &v_ex)
{
    return transform_all(v_ex, [](const expression &ex) {
        const auto &f = ex.fn();
        if (f.kind() == func_kind::pow && !f.args()[1].is_number()) {
            // NOTE: no constant folding for the log of a numerical base.
            return exp(f.args()[1] * detail::make_func(func_kind::log, {f.args()[0]}));
        }
        return ex;
    });
}

// Sums with negated terms -> subtractions (reference: src/math/sum.cpp:461-544).
std::vector<expression> sum_to_sub(const std::vector<expression> &v_ex)
{
    return transform_all(v_ex, [](const expression &ex) {
        const auto &fn = ex.fn();
        if (fn.kind() != func_kind::sum) {
            return ex;
        }

        auto new_args(fn.args());
        const auto fpart = [](const expression &arg) {
            if (is_kind(arg, func_kind::prod) && arg.fn().args().size() >= 2u && arg.fn().args()[0].is_number()) {
                return !is_negative_one(arg.fn().args()[0]);
            }
            return true;
        };
        const auto it = std::stable_partition(new_args.begin(), new_args.end(), fpart);

        if (it == new_args.end()) {
            return ex;
        }

        std::vector<expression> sub_args;
        for (auto cit = it; cit != new_args.end(); ++cit) {
            const auto &f = cit->fn();
            std::vector<expression> tmp_args(f.args().begin() + 1, f.args().end());
            sub_args.push_back(prod(std::move(tmp_args)));
        }

        auto st = sum(std::move(sub_args));

        if (it == new_args.begin()) {
            return prod({expression{-1.}, std::move(st)});
        }

        new_args.erase(it, new_args.end());
        auto mend = sum(std::move(new_args));
        return detail::sub(std::move(mend), std::move(st));
    });
}

// Re-organise a long associative function into nested invocations with at most 'split' arguments
// (reference: include/heyoka/detail/udf_split.hpp:49-100).
expression udf_split(const expression &e, func_kind k, std::uint32_t split)
{
    assert(split >= 2u);
    auto cur = e;
    while (true) {
        if (!is_kind(cur, k) || cur.fn().args().size() <= split) {
            return cur;
        }

        std::vector<expression> ret_seq, tmp;
        for (const auto &arg : cur.fn().args()) {
            tmp.push_back(arg);
            if (tmp.size() == split) {
                ret_seq.push_back(detail::make_func(k, std::move(tmp)));
                tmp.clear();
            }
        }
        if (!tmp.empty()) {
            if (tmp.size() == 1u) {
                ret_seq.push_back(std::move(tmp[0]));
            } else {
                ret_seq.push_back(detail::make_func(k, std::move(tmp)));
            }
        }
        cur = detail::make_func(k, std::move(ret_seq));
    }
}

// Reference: src/expression_basic.cpp:1177-1196 (split on 8).
std::vector<expression> split_sums_for_decompose(const std::vector<expression> &v_ex)
{
    return transform_all(v_ex, [](const expression &ex) { return udf_split(ex, func_kind::sum, 8); });
}

// Reference: src/expression_basic.cpp:1198-1213.
std::vector<expression> split_prods_for_decompose(const std::vector<expression> &v_ex, std::uint32_t split)
{
    return transform_all(v_ex, [split](const expression &ex) { return udf_split(ex, func_kind::prod, split); });
}

// sum({x**2, y**2, ...}) -> sum_sq({x, y, ...}) (reference: src/math/sum.cpp:385-455).
std::vector<expression> sums_to_sum_sqs_for_decompose(const std::vector<expression> &v_ex)
{
    return transform_all(v_ex, [](const expression &ex) {
        if (!is_kind(ex, func_kind::sum)) {
            return ex;
        }
        std::vector<expression> new_args;
        for (const auto &arg : ex.fn().args()) {
            if (is_kind(arg, func_kind::pow) && arg.fn().args()[1].is_number() && arg.fn().args()[1].num() == 2) {
                new_args.push_back(arg.fn().args()[0]);
            } else {
                return ex;
            }
        }
        return detail::sum_sq(std::move(new_args));
    });
}

// prod with pow(., -1) factors -> div (reference: src/math/prod.cpp:753-908).
std::vector<expression> prod_to_div_taylor_diff(const std::vector<expression> &v_ex)
{
    return transform_all(v_ex, [](const expression &ex) {
        if (!is_kind(ex, func_kind::prod)) {
            return ex;
        }

        // true -> keep in the numerator.
        const auto fpart = [](const expression &e) {
            if (!is_kind(e, func_kind::pow)) {
                return true;
            }
            const auto &expo = e.fn().args()[1];
            return !(expo.is_number() && expo.num() == -1);
        };

        auto new_args(ex.fn().args());
        const auto it = std::stable_partition(new_args.begin(), new_args.end(), fpart);
        if (it == new_args.end()) {
            return ex;
        }

        std::vector<expression> div_args;
        for (auto cit = it; cit != new_args.end(); ++cit) {
            const auto &f = cit->fn();
            div_args.push_back(pow(f.args()[0], expression{-f.args()[1].num()}));
        }
        auto divisor = prod(std::move(div_args));

        new_args.erase(it, new_args.end());
        auto num = prod(std::move(new_args));

        return detail::div(std::move(num), std::move(divisor));
    });
}

// Decomposition of a single function whose arguments have already been decomposed
// (reference: func_taylor_decompose_impl(), src/func.cpp:392-420; sin/cos custom
// decompositions src/math/sin.cpp:115-133, src/math/cos.cpp:116-134).
std::size_t func_taylor_decompose(expression f_ex, taylor_dc_t &dc)
{
    const auto &f = f_ex.fn();

    if (f.kind() == func_kind::sin || f.kind() == func_kind::cos) {
        const auto other = (f.kind() == func_kind::sin) ? func_kind::cos : func_kind::sin;
        // NOTE: the argument cannot be a number here (constant folding at construction), thus
        // building the partner function directly never folds.
        dc.emplace_back(detail::make_func(other, {f.args()[0]}), std::vector<std::uint32_t>{});
        dc.emplace_back(std::move(f_ex), std::vector<std::uint32_t>{});

        (dc.end() - 2)->second.push_back(static_cast<std::uint32_t>(dc.size() - 1u));
        (dc.end() - 1)->second.push_back(static_cast<std::uint32_t>(dc.size() - 2u));

        return dc.size() - 1u;
    }

    const auto u32 = [](std::size_t x) { return static_cast<std::uint32_t>(x); };
    const auto uvar = [&](std::size_t i) { return expression{uname(i)}; };
    const auto &arg = f.args().empty() ? f_ex : f.args()[0];

    switch (f.kind()) {
        case func_kind::tan:
        case func_kind::tanh:
        case func_kind::sigmoid: {
            // f(b) followed by its square, on which it depends
            // (src/math/tan.cpp:69-85, src/math/tanh.cpp:76-92, src/math/sigmoid.cpp:102-118).
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{});
            const auto i = dc.size() - 1u;
            dc.emplace_back(pow(uvar(i), expression{2.}), std::vector<std::uint32_t>{});
            dc[i].second.push_back(u32(i + 1u));
            return i;
        }
        case func_kind::sinh:
        case func_kind::cosh: {
            // Mutually-dependent pair, partner first (src/math/sinh.cpp:75-93, src/math/cosh.cpp:75-93).
            const auto other = (f.kind() == func_kind::sinh) ? func_kind::cosh : func_kind::sinh;
            dc.emplace_back(detail::make_func(other, {arg}), std::vector<std::uint32_t>{});
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{});
            (dc.end() - 2)->second.push_back(u32(dc.size() - 1u));
            (dc.end() - 1)->second.push_back(u32(dc.size() - 2u));
            return dc.size() - 1u;
        }
        case func_kind::asin:
        case func_kind::acos:
        case func_kind::asinh:
        case func_kind::acosh: {
            // b^2 -> (1 - b^2 | 1 + b^2 | b^2 - 1) -> sqrt -> f(b), which depends on the square root
            // (src/math/asin.cpp:77-108, acos.cpp:77-108, asinh.cpp:76-101, acosh.cpp:76-101).
            dc.emplace_back(pow(arg, expression{2.}), std::vector<std::uint32_t>{});
            const auto sq = uvar(dc.size() - 1u);
            if (f.kind() == func_kind::asin || f.kind() == func_kind::acos) {
                dc.emplace_back(detail::make_func(func_kind::sub, {expression{1.}, sq}), std::vector<std::uint32_t>{});
            } else if (f.kind() == func_kind::asinh) {
                dc.emplace_back(expression{1.} + sq, std::vector<std::uint32_t>{});
            } else {
                dc.emplace_back(sq - expression{1.}, std::vector<std::uint32_t>{});
            }
            dc.emplace_back(sqrt(uvar(dc.size() - 1u)), std::vector<std::uint32_t>{});
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{u32(dc.size() - 1u)});
            return dc.size() - 1u;
        }
        case func_kind::atan:
        case func_kind::atanh: {
            // b^2 -> f(b), which depends on it (src/math/atan.cpp:74-91, atanh.cpp:74-91).
            dc.emplace_back(pow(arg, expression{2.}), std::vector<std::uint32_t>{});
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{u32(dc.size() - 1u)});
            return dc.size() - 1u;
        }
        case func_kind::erf: {
            // b^2 -> -b^2 -> exp(-b^2) -> erf(b), which depends on the exponential (src/math/erf.cpp:81-105).
            dc.emplace_back(pow(arg, expression{2.}), std::vector<std::uint32_t>{});
            dc.emplace_back(-uvar(dc.size() - 1u), std::vector<std::uint32_t>{});
            dc.emplace_back(exp(uvar(dc.size() - 1u)), std::vector<std::uint32_t>{});
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{u32(dc.size() - 1u)});
            return dc.size() - 1u;
        }
        case func_kind::atan2: {
            // y^2 + x^2 -> atan2(y, x), which depends on it (src/math/atan2.cpp:92-108).
            dc.emplace_back(detail::sum_sq({f.args()[0], f.args()[1]}), std::vector<std::uint32_t>{});
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{u32(dc.size() - 1u)});
            return dc.size() - 1u;
        }
        case func_kind::kepE: {
            // E = kepE(e, M) -> sin(E) -> cos(E) -> e * cos(E); E depends on (e cos E, sin E), in this order, and
            // sin / cos on each other (src/math/kepE.cpp:100-135).
            const auto ecc = f.args()[0];
            dc.emplace_back(f_ex, std::vector<std::uint32_t>{});
            const auto iE = dc.size() - 1u;
            dc.emplace_back(sin(uvar(iE)), std::vector<std::uint32_t>{});
            dc.emplace_back(cos(uvar(iE)), std::vector<std::uint32_t>{});
            dc.emplace_back(ecc * uvar(iE + 2u), std::vector<std::uint32_t>{});
            dc[iE].second = {u32(iE + 3u), u32(iE + 1u)};
            dc[iE + 1u].second.push_back(u32(iE + 2u));
            dc[iE + 2u].second.push_back(u32(iE + 1u));
            return iE;
        }
        default:
            break;
    }

    const auto ret = dc.size();
    dc.emplace_back(std::move(f_ex), std::vector<std::uint32_t>{});
    return ret;
}

// Iterative post-order decomposition with a pointer cache
// (reference: expression_decompose_impl(), src/expression_decompose.cpp:43-210).
// NOTE: with taylor = false every function is appended as-is (decomposition of a compiled function,
// src/func.cpp:360-390: no partner functions / hidden dependencies).
std::optional<std::size_t> taylor_decompose(std::unordered_map<const void *, std::size_t> &func_map,
                                            const expression &e, taylor_dc_t &dc, bool taylor = true)
{
    std::vector<std::pair<const expression *, bool>> stack;
    std::vector<std::optional<std::optional<std::size_t>>> out_stack;

    stack.emplace_back(&e, false);

    while (!stack.empty()) {
        const auto [cur_ex, visited] = stack.back();
        stack.pop_back();

        if (cur_ex->is_func()) {
            const auto &f = cur_ex->fn();
            const auto *f_id = f.get_ptr();

            if (!visited) {
                if (const auto it = func_map.find(f_id); it != func_map.end()) {
                    out_stack.emplace_back(std::optional<std::size_t>{it->second});
                    continue;
                }
            }

            if (visited) {
                std::vector<expression> new_args;
                const auto n_args = f.args().size();
                new_args.reserve(n_args);
                for (std::size_t i = 0; i < n_args; ++i) {
                    assert(!out_stack.empty() && out_stack.back());
                    const auto opt_idx = *out_stack.back();
                    if (opt_idx) {
                        new_args.emplace_back(uname(*opt_idx));
                    } else {
                        new_args.push_back(f.args()[i]);
                    }
                    out_stack.pop_back();
                }

                std::size_t ret = 0;
                if (taylor) {
                    ret = func_taylor_decompose(expression{f.copy_with_new_args(std::move(new_args))}, dc);
                } else {
                    ret = dc.size();
                    dc.emplace_back(expression{f.copy_with_new_args(std::move(new_args))},
                                    std::vector<std::uint32_t>{});
                }
                if (ret == 0u || ret >= dc.size()) {
                    throw std::invalid_argument("Invalid value returned by the Taylor decomposition of a function");
                }

                func_map.emplace(f_id, ret);

                assert(!out_stack.empty() && !out_stack.back());
                out_stack.back().emplace(std::optional<std::size_t>{ret});
            } else {
                stack.emplace_back(cur_ex, true);
                for (const auto &ex : f.args()) {
                    stack.emplace_back(&ex, false);
                }
                out_stack.emplace_back();
            }
        } else {
            out_stack.emplace_back(std::optional<std::size_t>{});
        }
    }

    assert(out_stack.size() == 1u && out_stack.back());
    return *out_stack.back();
}

std::uint32_t remap_uidx(const std::unordered_map<std::string, std::string> &m, std::uint32_t idx)
{
    const auto it = m.find(uname(idx));
    assert(it != m.end());
    return uname_to_index(it->second);
}

// Common subexpression elimination (reference: taylor_decompose_cse(), src/taylor_01.cpp:315-443).
// NOTE: hidden deps are not considered when comparing subexpressions.
// NOTE: n_eq leading variable entries, n_outs trailing definitions (n_outs == n_eq in a Taylor decomposition;
// the decomposition of a compiled function has nvars / nouts instead, src/expression_cfunc.cpp:198-300).
taylor_dc_t taylor_decompose_cse(const taylor_dc_t &v_ex, std::size_t n_eq, std::size_t n_outs)
{
    assert(v_ex.size() >= n_eq + n_outs);

    taylor_dc_t new_dc;
    std::unordered_map<expression, std::size_t, expression_hash> ex_map;
    std::unordered_map<std::string, std::string> uvars_rename;
    ptr_ex_map cache;

    for (std::size_t i = 0; i < n_eq; ++i) {
        new_dc.push_back(v_ex[i]);
        uvars_rename.emplace(uname(i), uname(i));
    }

    for (auto i = n_eq; i < v_ex.size() - n_outs; ++i) {
        const auto &[orig_ex, orig_deps] = v_ex[i];

        auto new_ex = rename_variables(cache, orig_ex, uvars_rename);
        // NOTE: the cache is keyed on node identity, and each element of v_ex is a distinct
        // top-level node renamed with a *growing* map: clear the cache between elements.
        cache.clear();

        if (const auto it = ex_map.find(new_ex); it == ex_map.end()) {
            new_dc.emplace_back(new_ex, orig_deps);
            ex_map.emplace(std::move(new_ex), new_dc.size() - 1u);
            uvars_rename.emplace(uname(i), uname(new_dc.size() - 1u));
        } else {
            uvars_rename.emplace(uname(i), uname(it->second));
        }
    }

    for (auto i = v_ex.size() - n_outs; i < v_ex.size(); ++i) {
        const auto &[orig_ex, orig_deps] = v_ex[i];
        assert(!orig_ex.is_func() && orig_deps.empty());
        new_dc.emplace_back(rename_variables(cache, orig_ex, uvars_rename), orig_deps);
        cache.clear();
    }

    for (auto &[_, deps] : new_dc) {
        for (auto &idx : deps) {
            idx = remap_uidx(uvars_rename, idx);
        }
    }

    return new_dc;
}

// Breadth-first (Kahn) topological re-sort (reference: taylor_sort_dc(), src/taylor_01.cpp:454-645).
taylor_dc_t taylor_sort_dc(const taylor_dc_t &dc, std::size_t n_eq, std::size_t n_outs)
{
    assert(dc.size() >= n_eq + n_outs);

    // Vertex 0 = root, vertex i + 1 = u variable i.
    const auto n_vert = dc.size() - n_outs + 1u;
    std::vector<std::vector<std::size_t>> out_edges(n_vert);
    std::vector<std::size_t> in_degree(n_vert, 0);

    const auto add_edge = [&](std::size_t from, std::size_t to) {
        out_edges[from].push_back(to);
        ++in_degree[to];
    };

    for (std::size_t i = 0; i < n_eq; ++i) {
        add_edge(0, i + 1u);
    }

    for (auto i = n_eq; i < dc.size() - n_outs; ++i) {
        const auto vars = get_variables(dc[i].first);
        if (vars.empty()) {
            add_edge(0, i + 1u);
        } else {
            for (const auto &var : vars) {
                add_edge(uname_to_index(var) + 1u, i + 1u);
            }
        }
    }

    std::vector<std::size_t> v_idx;
    std::deque<std::size_t> tmp;
    tmp.push_back(0);

    while (!tmp.empty()) {
        const auto v = tmp.front();
        tmp.pop_front();
        v_idx.push_back(v);

        // NOTE: out edges processed in order of target vertex.
        auto targets = out_edges[v];
        std::sort(targets.begin(), targets.end());

        for (const auto t : targets) {
            assert(in_degree[t] > 0u);
            if (--in_degree[t] == 0u) {
                tmp.push_back(t);
            }
        }
    }

    assert(v_idx.size() == n_vert);

    for (std::size_t i = 0; i + 1u < v_idx.size(); ++i) {
        v_idx[i] = v_idx[i + 1u] - 1u;
    }
    v_idx.resize(dc.size());
    for (auto i = dc.size() - n_outs; i < dc.size(); ++i) {
        v_idx[i] = i;
    }

    std::unordered_map<std::string, std::string> remap;
    for (std::size_t i = 0; i < n_eq; ++i) {
        assert(v_idx[i] == i);
        remap.emplace(uname(i), uname(i));
    }
    for (auto i = n_eq; i < v_idx.size() - n_outs; ++i) {
        remap.emplace(uname(v_idx[i]), uname(i));
    }

    taylor_dc_t retval;
    retval.reserve(dc.size());
    for (const auto idx : v_idx) {
        const auto &[ex, deps] = dc[idx];
        ptr_ex_map cache;
        auto new_ex = rename_variables(cache, ex, remap);
        std::vector<std::uint32_t> new_deps;
        new_deps.reserve(deps.size());
        for (const auto d : deps) {
            new_deps.push_back(remap_uidx(remap, d));
        }
        retval.emplace_back(std::move(new_ex), std::move(new_deps));
    }

    return retval;
}

} // namespace

void validate_ode_sys(const std::vector<std::pair<expression, expression>> &sys)
{
    if (sys.empty()) {
        throw std::invalid_argument("Cannot integrate a system of zero equations");
    }

    std::vector<expression> sys_rhs;
    std::unordered_set<std::string> lhs_vars_set;

    for (const auto &[lhs, rhs] : sys) {
        sys_rhs.push_back(rhs);
        if (!lhs.is_variable()) {
            throw std::invalid_argument("Invalid system of differential equations detected: the "
                                        "left-hand side contains the expression '"
                                        + lhs.to_string() + "', which is not a variable");
        }
        const auto &name = lhs.var_name();
        if (name.rfind("__", 0) == 0) {
            throw std::invalid_argument("Invalid system of differential equations detected: the variable '" + name
                                        + "' appears in the left-hand side, but variables beginning with '__' are "
                                          "reserved for internal use");
        }
        if (!lhs_vars_set.insert(name).second) {
            throw std::invalid_argument("Invalid system of differential equations detected: the variable '" + name
                                        + "' appears in the left-hand side twice");
        }
    }

    for (const auto &var : get_variables(sys_rhs)) {
        if (lhs_vars_set.count(var) == 0u) {
            throw std::invalid_argument("Invalid system of differential equations detected: the variable '" + var
                                        + "' appears in the right-hand side but not in the left-hand side");
        }
    }
}

taylor_dc_t taylor_decompose_sys(const std::vector<std::pair<expression, expression>> &sys)
{
    std::vector<std::uint32_t> dummy;
    return taylor_decompose_sys(sys, {}, dummy);
}

// Reference: taylor_decompose_sys(sys, sv_funcs), src/taylor_01.cpp:848-1008. The extra functions are decomposed
// after the right-hand sides; here they ride along as additional trailing entries through CSE and sorting (which
// renumber them) and are stripped at the end.
taylor_dc_t taylor_decompose_sys(const std::vector<std::pair<expression, expression>> &sys,
                                 const std::vector<expression> &sv_funcs, std::vector<std::uint32_t> &sv_funcs_dc)
{
    const auto n_eq = sys.size();
    const auto n_sv = sv_funcs.size();

    std::unordered_map<std::string, std::string> repl_map;
    for (std::size_t i = 0; i < n_eq; ++i) {
        repl_map.emplace(sys[i].first.var_name(), uname(i));
    }

    std::vector<expression> all_ex;
    all_ex.reserve(n_eq);
    for (const auto &[lhs, rhs] : sys) {
        all_ex.push_back(rhs);
    }
    all_ex.insert(all_ex.end(), sv_funcs.begin(), sv_funcs.end());

    all_ex = pow_to_explog(all_ex);
    all_ex = sum_to_sub(all_ex);
    all_ex = split_sums_for_decompose(all_ex);
    all_ex = sums_to_sum_sqs_for_decompose(all_ex);
    all_ex = prod_to_div_taylor_diff(all_ex);
    all_ex = split_prods_for_decompose(all_ex, 2);

    all_ex = rename_variables(all_ex, repl_map);

    taylor_dc_t u_vars_defs;
    u_vars_defs.reserve(n_eq);
    for (const auto &[lhs, rhs] : sys) {
        u_vars_defs.emplace_back(lhs, std::vector<std::uint32_t>{});
    }

    taylor_dc_t outs;
    outs.reserve(n_eq);

    std::unordered_map<const void *, std::size_t> func_map;
    for (std::size_t i = 0; i < n_eq; ++i) {
        const auto &ex = all_ex[i];
        if (const auto dres = taylor_decompose(func_map, ex, u_vars_defs)) {
            outs.emplace_back(expression{uname(*dres)}, std::vector<std::uint32_t>{});
        } else {
            outs.emplace_back(ex, std::vector<std::uint32_t>{});
        }
    }

    for (std::size_t i = n_eq; i < all_ex.size(); ++i) {
        const auto &ex = all_ex[i];
        if (ex.is_variable()) {
            outs.emplace_back(ex, std::vector<std::uint32_t>{});
        } else if (const auto dres = taylor_decompose(func_map, ex, u_vars_defs)) {
            outs.emplace_back(expression{uname(*dres)}, std::vector<std::uint32_t>{});
        } else {
            throw std::invalid_argument("The extra functions in a Taylor decomposition cannot be constants or parameters");
        }
    }

    u_vars_defs.insert(u_vars_defs.end(), outs.begin(), outs.end());

    u_vars_defs = taylor_decompose_cse(u_vars_defs, n_eq, n_eq + n_sv);
    u_vars_defs = taylor_sort_dc(u_vars_defs, n_eq, n_eq + n_sv);

    sv_funcs_dc.clear();
    for (std::size_t i = u_vars_defs.size() - n_sv; i < u_vars_defs.size(); ++i) {
        sv_funcs_dc.push_back(uname_to_index(u_vars_defs[i].first.var_name()));
    }
    u_vars_defs.resize(u_vars_defs.size() - n_sv);

    // NOTE: sincos_combine_taylor() (src/detail/sincos_combine.cpp) only selects a fused
    // sin+cos evaluation at order 0: it does not change the structure of the decomposition. The
    // HIP emitter always evaluates sin/cos pairs with sincos().

    // Numbers -> num_identity (reference: src/taylor_01.cpp:788-804).
    for (auto i = n_eq; i < u_vars_defs.size() - n_eq; ++i) {
        auto &[ex, deps] = u_vars_defs[i];
        if (ex.is_number()) {
            ex = detail::num_identity(ex);
            deps.clear();
        }
    }

    return u_vars_defs;
}

// Reference: function_decompose(), src/expression_cfunc.cpp:723-900.
// NOTE: products are split in binary form (the reference uses groups of 8 here and evaluates them as a
// pairwise product tree): same values up to the association of the multiplications.
taylor_dc_t function_decompose(const std::vector<expression> &v_ex_, const std::vector<expression> &vars)
{
    if (v_ex_.empty()) {
        throw std::invalid_argument("Cannot decompose a function with no outputs");
    }

    std::unordered_set<std::string> var_set;
    for (const auto &ex : vars) {
        if (ex.is_variable()) {
            if (!var_set.emplace(ex.var_name()).second) {
                throw std::invalid_argument("Error in the decomposition of a function: the variable '" + ex.var_name()
                                            + "' appears in the user-provided list of variables twice");
            }
        } else {
            throw std::invalid_argument("Error in the decomposition of a function: the user-provided list of "
                                        "variables contains the expression '"
                                        + ex.to_string() + "', which is not a variable");
        }
    }
    for (const auto &var : get_variables(v_ex_)) {
        if (var_set.find(var) == var_set.end()) {
            throw std::invalid_argument("Error in the decomposition of a function: the variable '" + var
                                        + "' appears in the function but not in the user-provided list of variables");
        }
    }

    const auto nvars = vars.size();
    const auto nouts = v_ex_.size();

    std::unordered_map<std::string, std::string> repl_map;
    for (std::size_t i = 0; i < nvars; ++i) {
        repl_map.emplace(vars[i].var_name(), uname(i));
    }

    auto v_ex = sum_to_sub(v_ex_);
    v_ex = split_sums_for_decompose(v_ex);
    v_ex = sums_to_sum_sqs_for_decompose(v_ex);
    v_ex = prod_to_div_taylor_diff(v_ex);
    v_ex = split_prods_for_decompose(v_ex, 2);
    v_ex = rename_variables(v_ex, repl_map);

    taylor_dc_t ret;
    for (const auto &var : vars) {
        ret.emplace_back(var, std::vector<std::uint32_t>{});
    }
    taylor_dc_t outs;
    std::unordered_map<const void *, std::size_t> func_map;
    for (const auto &ex : v_ex) {
        if (const auto dres = taylor_decompose(func_map, ex, ret, false)) {
            outs.emplace_back(expression{uname(*dres)}, std::vector<std::uint32_t>{});
        } else {
            outs.emplace_back(ex, std::vector<std::uint32_t>{});
        }
    }
    ret.insert(ret.end(), outs.begin(), outs.end());

    ret = taylor_decompose_cse(ret, nvars, nouts);
    ret = taylor_sort_dc(ret, nvars, nouts);

    for (auto i = nvars; i < ret.size() - nouts; ++i) {
        auto &[ex, deps] = ret[i];
        if (ex.is_number()) {
            ex = detail::num_identity(ex);
        }
    }
    return ret;
}

namespace
{

operand make_operand(const expression &e)
{
    operand op;
    if (e.is_number()) {
        op.type = operand::kind::num;
        op.value = e.num();
    } else if (e.is_param()) {
        op.type = operand::kind::par;
        op.idx = e.par_idx();
    } else if (e.is_variable()) {
        op.type = operand::kind::uvar;
        op.idx = uname_to_index(e.var_name());
    } else {
        throw std::invalid_argument("Invalid operand in a Taylor decomposition: '" + e.to_string() + "'");
    }
    return op;
}

} // namespace

taylor_program make_program(const taylor_dc_t &dc, std::uint32_t n_eq, std::uint32_t n_outs_)
{
    const auto n_outs = (n_outs_ == std::numeric_limits<std::uint32_t>::max()) ? n_eq : n_outs_;
    assert(dc.size() >= static_cast<std::size_t>(n_eq) + n_outs);

    taylor_program prog;
    prog.n_eq = n_eq;
    prog.n_u = static_cast<std::uint32_t>(dc.size() - n_outs);

    std::vector<expression> all;
    for (std::size_t i = n_eq; i < dc.size(); ++i) {
        all.push_back(dc[i].first);
    }
    prog.n_par = get_param_size(all);
    prog.time_dependent = is_time_dependent(all);

    for (std::size_t i = n_eq; i < prog.n_u; ++i) {
        const auto &[ex, deps] = dc[i];
        if (!ex.is_func()) {
            throw std::invalid_argument("Invalid Taylor decomposition: the definition of a u variable is not a function");
        }
        dc_node n;
        n.kind = ex.fn().kind();
        for (const auto &a : ex.fn().args()) {
            n.args.push_back(make_operand(a));
            if (n.args.back().type == operand::kind::uvar && n.args.back().idx >= i) {
                throw std::invalid_argument("Invalid Taylor decomposition: forward reference to a u variable");
            }
        }
        n.deps = deps;
        prog.nodes.push_back(std::move(n));
    }

    for (std::size_t i = prog.n_u; i < dc.size(); ++i) {
        prog.sv_defs.push_back(make_operand(dc[i].first));
    }

    return prog;
}

std::uint32_t taylor_order_from_tol(double tol)
{
    auto order_f = std::ceil(-std::log(tol) / 2 + 1);
    if (!std::isfinite(order_f)) {
        throw std::invalid_argument(
            "The computation of the Taylor order in an adaptive Taylor stepper produced a non-finite value");
    }
    order_f = std::max(2., order_f);
    if (order_f > static_cast<double>(std::numeric_limits<std::uint32_t>::max())) {
        throw std::overflow_error("The computation of the Taylor order in an adaptive Taylor stepper resulted "
                                  "in an overflow condition");
    }
    return static_cast<std::uint32_t>(order_f);
}

} // namespace heyoka_amd
