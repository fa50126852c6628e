// Taylor decomposition implementation. See decompose.hpp for the reference citations.
#include "decompose.hpp"
#include "node_rule.hpp"

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

expression uvar(std::size_t i)
{
    return expression{uname(i)};
}

bool has_kind(const expression &e, func_kind k)
{
    return e.is_func() && e.fn().kind() == k;
}

// ---------------------------------------------------------------------------------------------------------------------
// Rewrites ahead of the decomposition. Each of them is a LOCAL rule - a function node whose arguments have been
// rewritten already goes in, its replacement comes out - applied to every function node of the right-hand sides,
// arguments first, shared nodes once (rewrite_everywhere()). What the rules have to produce is fixed by the reference
// (the numbering of the u variables depends on it); how they are written is not.
// ---------------------------------------------------------------------------------------------------------------------
using local_rule = std::function<expression(const expression &)>;

std::vector<expression> rewrite_everywhere(const std::vector<expression> &roots, const local_rule &rule)
{
    ptr_ex_map done;
    std::vector<expression> out;
    out.reserve(roots.size());
    for (const auto &r : roots) {
        out.push_back(traverse_transform_nodes(done, r, {}, rule));
    }
    return out;
}

// pow(x, y) with a non-numerical exponent: exp(y * log(x)) (src/taylor_01.cpp:806-840; the logarithm of a numerical
// base is not folded).
expression rule_pow_to_explog(const expression &ex)
{
    const auto &f = ex.fn();
    if (f.kind() != func_kind::pow || f.args()[1].is_number()) {
        return ex;
    }
    return exp(f.args()[1] * detail::make_func(func_kind::log, {f.args()[0]}));
}

// A sum in which some terms are products led by the number -1: (sum of the other terms) - (sum of those terms without
// their -1); all terms negative: -1 * (sum of them) (src/math/sum.cpp:461-544).
expression rule_sum_to_sub(const expression &ex)
{
    if (!has_kind(ex, func_kind::sum)) {
        return ex;
    }
    std::vector<expression> plus, minus;
    for (const auto &term : ex.fn().args()) {
        const bool negated = has_kind(term, func_kind::prod) && term.fn().args().size() >= 2u
                             && term.fn().args()[0].is_number() && term.fn().args()[0].num() == -1;
        if (negated) {
            const auto &pa = term.fn().args();
            minus.push_back(prod(std::vector<expression>(pa.begin() + 1, pa.end())));
        } else {
            plus.push_back(term);
        }
    }
    if (minus.empty()) {
        return ex;
    }
    auto subtrahend = sum(std::move(minus));
    if (plus.empty()) {
        return prod({expression{-1.}, std::move(subtrahend)});
    }
    return detail::sub(sum(std::move(plus)), std::move(subtrahend));
}

// A function of kind k with more than `width` arguments: the arguments in runs of `width` become functions of their own
// (a last run of one argument stays as it is), repeated on the result until it fits
// (include/heyoka/detail/udf_split.hpp:49-100).
expression in_runs_of(const expression &ex, func_kind k, std::size_t width)
{
    auto cur = ex;
    while (has_kind(cur, k) && cur.fn().args().size() > width) {
        const auto &args = cur.fn().args();
        std::vector<expression> grouped;
        for (std::size_t first = 0; first < args.size(); first += width) {
            const auto last = std::min(args.size(), first + width);
            if (last - first == 1u) {
                grouped.push_back(args[first]);
            } else {
                grouped.push_back(detail::make_func(
                    k, std::vector<expression>(args.begin() + static_cast<std::ptrdiff_t>(first),
                                               args.begin() + static_cast<std::ptrdiff_t>(last))));
            }
        }
        cur = detail::make_func(k, std::move(grouped));
    }
    return cur;
}

// A sum whose terms are ALL squares: sum_sq of the bases (src/math/sum.cpp:385-455).
expression rule_sum_of_squares(const expression &ex)
{
    if (!has_kind(ex, func_kind::sum)) {
        return ex;
    }
    std::vector<expression> bases;
    for (const auto &term : ex.fn().args()) {
        const bool square = has_kind(term, func_kind::pow) && term.fn().args()[1].is_number() && term.fn().args()[1].num() == 2;
        if (!square) {
            return ex;
        }
        bases.push_back(term.fn().args()[0]);
    }
    return detail::sum_sq(std::move(bases));
}

// A product with factors pow(x, -1): (product of the other factors) / (product of the x) (src/math/prod.cpp:753-908).
expression rule_prod_to_div(const expression &ex)
{
    if (!has_kind(ex, func_kind::prod)) {
        return ex;
    }
    std::vector<expression> above, below;
    for (const auto &factor : ex.fn().args()) {
        const bool reciprocal = has_kind(factor, func_kind::pow) && factor.fn().args()[1].is_number()
                                && factor.fn().args()[1].num() == -1;
        if (reciprocal) {
            // pow(x, -(-1)): folds to x.
            below.push_back(pow(factor.fn().args()[0], expression{-factor.fn().args()[1].num()}));
        } else {
            above.push_back(factor);
        }
    }
    if (below.empty()) {
        return ex;
    }
    return detail::div(prod(std::move(above)), prod(std::move(below)));
}

std::vector<expression> rewrites_for_decomposition(std::vector<expression> v, bool with_explog)
{
    if (with_explog) {
        v = rewrite_everywhere(v, rule_pow_to_explog);
    }
    v = rewrite_everywhere(v, rule_sum_to_sub);
    // Sums in runs of 8 (src/expression_basic.cpp:1177-1196).
    v = rewrite_everywhere(v, [](const expression &e) { return in_runs_of(e, func_kind::sum, 8); });
    v = rewrite_everywhere(v, rule_sum_of_squares);
    v = rewrite_everywhere(v, rule_prod_to_div);
    // Binary products (src/expression_basic.cpp:1198-1213; the derivative rule of a product takes two factors).
    v = rewrite_everywhere(v, [](const expression &e) { return in_runs_of(e, func_kind::prod, 2); });
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// One function whose arguments are leaves (u variables, numbers, parameters) becomes one entry of the decomposition -
// plus, for the functions whose derivative rule reads other functions of the same argument ("hidden dependencies"),
// the entries of those. The ORDER of the entries and of the dependency lists is the reference's (it fixes the numbering
// and what a rule finds in deps[0], deps[1], ...): func_taylor_decompose_impl(), src/func.cpp:392-420, and the
// taylor_decompose() members of src/math/*.cpp cited below. Written as a table of "plans": each plan lists the entries
// to append in terms of `self` (the function itself), `arg` (its first argument), earlier entries of the plan, and the
// dependencies between them.
// ---------------------------------------------------------------------------------------------------------------------
struct plan_entry {
    // Builds the entry from: the function itself, and a resolver from plan positions to u variables.
    std::function<expression(const expression &self, const std::function<expression(int)> &at)> make;
    std::vector<int> deps; // plan positions this entry depends on
};
struct plan {
    std::vector<plan_entry> entries;
    int result = 0; // plan position of the function itself
};

const plan *plan_for(func_kind k)
{
    using E = const expression &;
    using At = const std::function<expression(int)> &;
    const auto arg0 = [](E self) -> const expression & { return self.fn().args()[0]; };
    static const auto self_entry = [](std::vector<int> deps) {
        return plan_entry{[](E self, At) { return self; }, std::move(deps)};
    };
    static const auto partner_entry = [](func_kind other, std::vector<int> deps) {
        return plan_entry{[other](E self, At) { return detail::make_func(other, {self.fn().args()[0]}); }, std::move(deps)};
    };
    static const auto square_of = [](int pos) {
        return plan_entry{[pos](E, At at) { return pow(at(pos), expression{2.}); }, {}};
    };
    static const auto square_of_arg = plan_entry{[](E self, At) { return pow(self.fn().args()[0], expression{2.}); }, {}};
    (void)arg0;

    // sin / cos and sinh / cosh: the partner first, each depends on the other (src/math/sin.cpp:115-133, cos.cpp:116-134,
    // sinh.cpp:75-93, cosh.cpp:75-93).
    static const plan p_sin{{partner_entry(func_kind::cos, {1}), self_entry({0})}, 1};
    static const plan p_cos{{partner_entry(func_kind::sin, {1}), self_entry({0})}, 1};
    static const plan p_sinh{{partner_entry(func_kind::cosh, {1}), self_entry({0})}, 1};
    static const plan p_cosh{{partner_entry(func_kind::sinh, {1}), self_entry({0})}, 1};
    // tan, tanh, sigmoid: the function, then its square, which it reads (src/math/tan.cpp:69-85, tanh.cpp:76-92,
    // sigmoid.cpp:102-118).
    static const plan p_self_then_square{{self_entry({1}), square_of(0)}, 0};
    // atan, atanh: the square of the argument, then the function (src/math/atan.cpp:74-91, atanh.cpp:74-91).
    static const plan p_argsq_then_self{{square_of_arg, self_entry({0})}, 1};
    // asin, acos: b^2, 1 - b^2, its square root, the function reads the root (src/math/asin.cpp:77-108, acos.cpp:77-108);
    // asinh: 1 + b^2; acosh: b^2 - 1 (asinh.cpp:76-101, acosh.cpp:76-101).
    static const auto root_chain = [](int variant) {
        return plan{{square_of_arg,
                     plan_entry{[variant](E, At at) {
                                    if (variant == 0) {
                                        return detail::make_func(func_kind::sub, {expression{1.}, at(0)});
                                    }
                                    return variant == 1 ? expression{1.} + at(0) : at(0) - expression{1.};
                                },
                                {}},
                     plan_entry{[](E, At at) { return sqrt(at(1)); }, {}}, self_entry({2})},
                    3};
    };
    static const plan p_one_minus = root_chain(0), p_one_plus = root_chain(1), p_minus_one = root_chain(2);
    // erf: b^2, -b^2, exp(-b^2), the function reads the exponential (src/math/erf.cpp:81-105).
    static const plan p_erf{{square_of_arg, plan_entry{[](E, At at) { return -at(0); }, {}},
                             plan_entry{[](E, At at) { return exp(at(1)); }, {}}, self_entry({2})},
                            3};
    // atan2(y, x): y^2 + x^2, then the function (src/math/atan2.cpp:92-108).
    static const plan p_atan2{
        {plan_entry{[](E self, At) { return detail::sum_sq({self.fn().args()[0], self.fn().args()[1]}); }, {}}, self_entry({0})}, 1};
    // E = kepE(e, M): E, sin E, cos E, e cos E; E reads (e cos E, sin E), sin and cos each other (src/math/kepE.cpp:100-135).
    static const plan p_kepE{{self_entry({3, 1}), plan_entry{[](E, At at) { return sin(at(0)); }, {2}},
                              plan_entry{[](E, At at) { return cos(at(0)); }, {1}},
                              plan_entry{[](E self, At at) { return self.fn().args()[0] * at(2); }, {}}},
                             0};

    switch (k) {
        case func_kind::sin:
            return &p_sin;
        case func_kind::cos:
            return &p_cos;
        case func_kind::sinh:
            return &p_sinh;
        case func_kind::cosh:
            return &p_cosh;
        case func_kind::tan:
        case func_kind::tanh:
        case func_kind::sigmoid:
            return &p_self_then_square;
        case func_kind::atan:
        case func_kind::atanh:
            return &p_argsq_then_self;
        case func_kind::asin:
        case func_kind::acos:
            return &p_one_minus;
        case func_kind::asinh:
            return &p_one_plus;
        case func_kind::acosh:
            return &p_minus_one;
        case func_kind::erf:
            return &p_erf;
        case func_kind::atan2:
            return &p_atan2;
        case func_kind::kepE:
            return &p_kepE;
        default:
            return nullptr;
    }
}

// Appends the entries of f (arguments: leaves) to dc; returns the index of the entry of f itself.
std::size_t append_function(const expression &f_ex, taylor_dc_t &dc)
{
    const auto base = dc.size();
    const auto at = [base](int pos) { return uvar(base + static_cast<std::size_t>(pos)); };

    if (f_ex.fn().kind() == func_kind::custom) {
        // A function defined through the registry (node_rule.hpp): the function, then its hidden definitions in the order
        // the rule lists them.
        const auto &rule = get_node_rule(f_ex.fn().rule());
        dc.emplace_back(f_ex, std::vector<std::uint32_t>{});
        if (rule.decompose) {
            const auto hidden = [base](std::uint32_t j) { return uvar(base + 1u + j); };
            const auto defs = rule.decompose(uvar(base), f_ex.fn().args(), hidden);
            for (const auto &d : defs) {
                std::vector<std::uint32_t> deps;
                for (const auto j : d.deps) {
                    if (j >= defs.size()) {
                        throw std::invalid_argument("Invalid hidden dependency in the decomposition of '" + rule.name + "'");
                    }
                    deps.push_back(static_cast<std::uint32_t>(base + 1u + j));
                }
                // (A hidden definition is ONE elementary function of leaves: a rule which builds it with the folding operators
                // can end up with a number or a bare variable - 0 * x, 1 * x.)
                bool leaves_only = d.ex.is_func();
                for (std::size_t q = 0; leaves_only && q < d.ex.fn().args().size(); ++q) {
                    leaves_only = !d.ex.fn().args()[q].is_func();
                }
                if (!leaves_only) {
                    throw std::invalid_argument("The decomposition of the node rule '" + rule.name
                                                + "' returned a hidden definition which is not one function of variables, numbers and "
                                                  "parameters (folded by an operator? build it with make_func())");
                }
                dc.emplace_back(d.ex, std::move(deps));
            }
            for (const auto j : rule.deps) {
                if (j >= defs.size()) {
                    throw std::invalid_argument("Invalid hidden dependency in the node rule '" + rule.name + "'");
                }
                dc[base].second.push_back(static_cast<std::uint32_t>(base + 1u + j));
            }
        } else if (!rule.deps.empty()) {
            throw std::invalid_argument("The node rule '" + rule.name + "' lists hidden dependencies but cannot decompose");
        }
        return base;
    }

    const auto *pl = plan_for(f_ex.fn().kind());
    if (pl == nullptr) {
        dc.emplace_back(f_ex, std::vector<std::uint32_t>{});
        return base;
    }
    for (const auto &pe : pl->entries) {
        std::vector<std::uint32_t> deps;
        for (const auto d : pe.deps) {
            deps.push_back(static_cast<std::uint32_t>(base + static_cast<std::size_t>(d)));
        }
        dc.emplace_back(pe.make(f_ex, at), std::move(deps));
    }
    return base + static_cast<std::size_t>(pl->result);
}

// Decomposition of one expression: every function node (arguments first, shared nodes once - `seen` maps the identity
// of a node to its u variable across the expressions of a system) is re-stated over u variables and appended
// (expression_decompose_impl(), src/expression_decompose.cpp:43-210). Returns the u variable holding e, or nothing if e
// is a leaf. plain = true: compiled functions - no hidden dependencies (src/func.cpp:360-390).
std::optional<std::size_t> decompose_expression(std::unordered_map<const void *, std::size_t> &seen, const expression &e,
                                                taylor_dc_t &dc, bool plain = false)
{
    if (!e.is_func()) {
        return std::nullopt;
    }
    const auto todo = function_nodes_postorder(e, [&seen](const void *id) { return seen.count(id) != 0u; });
    for (const auto *node : todo) {
        const auto &f = node->fn();
        std::vector<expression> leaves;
        leaves.reserve(f.args().size());
        for (const auto &a : f.args()) {
            leaves.push_back(a.is_func() ? uvar(seen.at(a.fn().get_ptr())) : a);
        }
        const expression restated{f.copy_with_new_args(std::move(leaves))};
        std::size_t idx = dc.size();
        if (plain) {
            dc.emplace_back(restated, std::vector<std::uint32_t>{});
        } else {
            idx = append_function(restated, dc);
        }
        if (idx == 0u || idx >= dc.size()) {
            throw std::invalid_argument("Invalid value returned by the Taylor decomposition of a function");
        }
        seen.emplace(f.get_ptr(), idx);
    }
    return seen.at(e.fn().get_ptr());
}

// The u index named by a leaf, if it is a u variable.
std::optional<std::uint32_t> leaf_uindex(const expression &a)
{
    if (a.is_variable()) {
        const auto &n = a.var_name();
        if (n.size() > 2u && n[0] == 'u' && n[1] == '_' && n.find_first_not_of("0123456789", 2) == std::string::npos) {
            return uname_to_index(n);
        }
    }
    return std::nullopt;
}

// An entry (one function of leaves, or a leaf for the trailing definitions) with its u variables renumbered.
expression renumbered(const expression &ex, const std::vector<std::uint32_t> &new_index)
{
    const auto leaf = [&new_index](const expression &a) {
        const auto u = leaf_uindex(a);
        return u ? uvar(new_index[*u]) : a;
    };
    if (!ex.is_func()) {
        return leaf(ex);
    }
    std::vector<expression> args;
    args.reserve(ex.fn().args().size());
    for (const auto &a : ex.fn().args()) {
        assert(!a.is_func());
        args.push_back(leaf(a));
    }
    return expression{ex.fn().copy_with_new_args(std::move(args))};
}

// Elimination of repeated entries (taylor_decompose_cse(), src/taylor_01.cpp:315-443): walking the entries in order with
// the renumbering built so far, an entry equal to an earlier one (same function, same arguments - hidden dependencies do
// not take part in the comparison) is dropped and its index redirected to the first occurrence.
// n_lead leading variable entries, n_outs trailing definitions.
taylor_dc_t merge_repeated_entries(const taylor_dc_t &dc, std::size_t n_lead, std::size_t n_outs)
{
    assert(dc.size() >= n_lead + n_outs);
    const auto n_mid_end = dc.size() - n_outs;

    std::vector<std::uint32_t> new_index(dc.size(), 0u);
    std::unordered_map<expression, std::uint32_t, expression_hash> first_seen;
    taylor_dc_t out;
    out.reserve(dc.size());

    for (std::size_t i = 0; i < n_lead; ++i) {
        new_index[i] = static_cast<std::uint32_t>(i);
        out.push_back(dc[i]);
    }
    for (auto i = n_lead; i < n_mid_end; ++i) {
        auto ex = renumbered(dc[i].first, new_index);
        const auto pos = first_seen.find(ex);
        if (pos != first_seen.end()) {
            new_index[i] = pos->second;
            continue;
        }
        new_index[i] = static_cast<std::uint32_t>(out.size());
        first_seen.emplace(ex, new_index[i]);
        // (The dependency lists still carry the old numbering: fixed below, once it is complete.)
        out.emplace_back(std::move(ex), dc[i].second);
    }
    for (auto i = n_mid_end; i < dc.size(); ++i) {
        assert(dc[i].second.empty());
        out.emplace_back(renumbered(dc[i].first, new_index), std::vector<std::uint32_t>{});
    }
    for (auto &entry : out) {
        for (auto &d : entry.second) {
            d = new_index[d];
        }
    }
    return out;
}

// Breadth-first re-ordering (taylor_sort_dc(), src/taylor_01.cpp:454-645): level by level from the state variables, an
// entry becomes ready when all the u variables among its arguments have been placed; ready entries are placed in the order
// in which they became ready, and the entries released by one placement in increasing index. Entries without u variables
// among their arguments are released by a virtual root, together with the state variables.
taylor_dc_t breadth_first_order(const taylor_dc_t &dc, std::size_t n_lead, std::size_t n_outs)
{
    assert(dc.size() >= n_lead + n_outs);
    const auto n_mid_end = dc.size() - n_outs;

    // users[v]: the entries reading u variable v (each once, increasing); users of the root at index n_mid_end.
    std::vector<std::vector<std::uint32_t>> users(n_mid_end + 1u);
    std::vector<std::uint32_t> missing(n_mid_end, 0u);
    for (std::size_t i = 0; i < n_lead; ++i) {
        users[n_mid_end].push_back(static_cast<std::uint32_t>(i));
        missing[i] = 1;
    }
    for (auto i = n_lead; i < n_mid_end; ++i) {
        std::set<std::uint32_t> reads;
        for (const auto &a : dc[i].first.fn().args()) {
            if (const auto u = leaf_uindex(a)) {
                reads.insert(*u);
            }
        }
        if (reads.empty()) {
            users[n_mid_end].push_back(static_cast<std::uint32_t>(i));
            missing[i] = 1;
        } else {
            for (const auto u : reads) {
                users[u].push_back(static_cast<std::uint32_t>(i));
            }
            missing[i] = static_cast<std::uint32_t>(reads.size());
        }
    }

    std::vector<std::uint32_t> placed; // old indices in their new order
    placed.reserve(n_mid_end);
    std::deque<std::size_t> ready{n_mid_end};
    while (!ready.empty()) {
        const auto v = ready.front();
        ready.pop_front();
        if (v != n_mid_end) {
            placed.push_back(static_cast<std::uint32_t>(v));
        }
        for (const auto t : users[v]) {
            assert(missing[t] > 0u);
            if (--missing[t] == 0u) {
                ready.push_back(t);
            }
        }
    }
    assert(placed.size() == n_mid_end);

    std::vector<std::uint32_t> new_index(dc.size(), 0u);
    for (std::size_t pos = 0; pos < placed.size(); ++pos) {
        new_index[placed[pos]] = static_cast<std::uint32_t>(pos);
    }
    for (auto i = n_mid_end; i < dc.size(); ++i) {
        new_index[i] = static_cast<std::uint32_t>(i);
    }

    taylor_dc_t out;
    out.reserve(dc.size());
    const auto emit = [&](std::size_t old) {
        std::vector<std::uint32_t> deps;
        deps.reserve(dc[old].second.size());
        for (const auto d : dc[old].second) {
            deps.push_back(new_index[d]);
        }
        out.emplace_back(renumbered(dc[old].first, new_index), std::move(deps));
    };
    for (const auto old : placed) {
        emit(old);
    }
    for (auto i = n_mid_end; i < dc.size(); ++i) {
        emit(i);
    }
    return out;
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
    std::vector<std::uint32_t> unused;
    return taylor_decompose_sys(sys, {}, unused);
}

namespace
{

// Numerical definitions of u variables become num_identity() functions (src/taylor_01.cpp:788-804).
void wrap_numbers(taylor_dc_t &dc, std::size_t first, std::size_t last)
{
    for (auto i = first; i < last; ++i) {
        if (dc[i].first.is_number()) {
            dc[i].first = detail::num_identity(dc[i].first);
            dc[i].second.clear();
        }
    }
}

} // namespace

// Reference: taylor_decompose_sys(sys, sv_funcs), src/taylor_01.cpp:848-1008. Layout of the result: the state variables,
// one entry per elementary function, the definitions of the derivatives of the state variables. The extra functions of
// the state (event equations, ...) are decomposed behind the right-hand sides and travel as additional trailing
// definitions through the merging and re-ordering passes, which renumber them; they are cut off at the end.
taylor_dc_t taylor_decompose_sys(const std::vector<std::pair<expression, expression>> &sys,
                                 const std::vector<expression> &sv_funcs, std::vector<std::uint32_t> &sv_funcs_dc)
{
    const auto n_state = sys.size();
    const auto n_extra = sv_funcs.size();

    // Right-hand sides, then the extra functions: rewritten together (shared subexpressions stay shared) and re-stated
    // over u_0 .. u_{n-1} in place of the state variables.
    std::vector<expression> work;
    work.reserve(n_state + n_extra);
    std::unordered_map<std::string, std::string> state_to_u;
    for (std::size_t i = 0; i < n_state; ++i) {
        work.push_back(sys[i].second);
        state_to_u.emplace(sys[i].first.var_name(), uname(i));
    }
    work.insert(work.end(), sv_funcs.begin(), sv_funcs.end());
    work = rename_variables(rewrites_for_decomposition(std::move(work), true), state_to_u);

    taylor_dc_t dc;
    for (const auto &eq : sys) {
        dc.emplace_back(eq.first, std::vector<std::uint32_t>{});
    }

    // The trailing definitions: where each right-hand side / extra function ended up.
    taylor_dc_t tail;
    std::unordered_map<const void *, std::size_t> seen;
    for (std::size_t i = 0; i < work.size(); ++i) {
        const bool extra = i >= n_state;
        if (extra && work[i].is_variable()) {
            tail.emplace_back(work[i], std::vector<std::uint32_t>{});
            continue;
        }
        const auto where = decompose_expression(seen, work[i], dc);
        if (where) {
            tail.emplace_back(uvar(*where), std::vector<std::uint32_t>{});
        } else if (extra) {
            throw std::invalid_argument("The extra functions in a Taylor decomposition cannot be constants or parameters");
        } else {
            tail.emplace_back(work[i], std::vector<std::uint32_t>{});
        }
    }
    dc.insert(dc.end(), tail.begin(), tail.end());

    dc = merge_repeated_entries(dc, n_state, n_state + n_extra);
    dc = breadth_first_order(dc, n_state, n_state + n_extra);

    sv_funcs_dc.clear();
    for (auto i = dc.size() - n_extra; i < dc.size(); ++i) {
        sv_funcs_dc.push_back(uname_to_index(dc[i].first.var_name()));
    }
    dc.resize(dc.size() - n_extra);

    // NOTE: sincos_combine_taylor() (src/detail/sincos_combine.cpp) only selects a fused sin + cos evaluation at order 0:
    // it does not change the structure of the decomposition. The HIP emitters always evaluate sin / cos pairs together.
    wrap_numbers(dc, n_state, dc.size() - n_state);
    return dc;
}

// Reference: function_decompose(), src/expression_cfunc.cpp:723-900.
// NOTE: products are split in binary form (the reference uses groups of 8 here and evaluates them as a
// pairwise product tree): same values up to the association of the multiplications.
taylor_dc_t function_decompose(const std::vector<expression> &fn, const std::vector<expression> &vars)
{
    if (fn.empty()) {
        throw std::invalid_argument("Cannot decompose a function with no outputs");
    }

    std::unordered_map<std::string, std::string> var_to_u;
    for (const auto &v : vars) {
        if (!v.is_variable()) {
            throw std::invalid_argument("Error in the decomposition of a function: the user-provided list of "
                                        "variables contains the expression '"
                                        + v.to_string() + "', which is not a variable");
        }
        if (!var_to_u.emplace(v.var_name(), uname(var_to_u.size())).second) {
            throw std::invalid_argument("Error in the decomposition of a function: the variable '" + v.var_name()
                                        + "' appears in the user-provided list of variables twice");
        }
    }
    for (const auto &name : get_variables(fn)) {
        if (var_to_u.count(name) == 0u) {
            throw std::invalid_argument("Error in the decomposition of a function: the variable '" + name
                                        + "' appears in the function but not in the user-provided list of variables");
        }
    }

    const auto n_vars = vars.size(), n_outs = fn.size();
    const auto work = rename_variables(rewrites_for_decomposition(fn, false), var_to_u);

    taylor_dc_t dc;
    for (const auto &v : vars) {
        dc.emplace_back(v, std::vector<std::uint32_t>{});
    }
    taylor_dc_t tail;
    std::unordered_map<const void *, std::size_t> seen;
    for (const auto &ex : work) {
        const auto where = decompose_expression(seen, ex, dc, true);
        tail.emplace_back(where ? uvar(*where) : ex, std::vector<std::uint32_t>{});
    }
    dc.insert(dc.end(), tail.begin(), tail.end());

    dc = merge_repeated_entries(dc, n_vars, n_outs);
    dc = breadth_first_order(dc, n_vars, n_outs);
    for (auto i = n_vars; i < dc.size() - n_outs; ++i) {
        if (dc[i].first.is_number()) {
            dc[i].first = detail::num_identity(dc[i].first);
        }
    }
    return dc;
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
        n.rule = ex.fn().rule();
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
