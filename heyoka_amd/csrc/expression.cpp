// Expression system implementation. See expression.hpp for the reference citations.
#include "expression.hpp"
#include "node_rule.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <optional>
#include <set>
#include <sstream>
#include <iomanip>

namespace heyoka_amd
{

const char *func_kind_name(func_kind k)
{
    switch (k) {
        case func_kind::sum:
            return "sum";
        case func_kind::prod:
            return "prod";
        case func_kind::pow:
            return "pow";
        case func_kind::sub:
            return "sub";
        case func_kind::div:
            return "div";
        case func_kind::sum_sq:
            return "sum_sq";
        case func_kind::sin:
            return "sin";
        case func_kind::cos:
            return "cos";
        case func_kind::exp:
            return "exp";
        case func_kind::log:
            return "log";
        case func_kind::time:
            return "time";
        case func_kind::num_identity:
            return "num_identity";
        case func_kind::tan:
            return "tan";
        case func_kind::tanh:
            return "tanh";
        case func_kind::sinh:
            return "sinh";
        case func_kind::cosh:
            return "cosh";
        case func_kind::asin:
            return "asin";
        case func_kind::acos:
            return "acos";
        case func_kind::atan:
            return "atan";
        case func_kind::atan2:
            return "atan2";
        case func_kind::kepE:
            return "kepE";
        case func_kind::relu:
            return "relu";
        case func_kind::relup:
            return "relup";
        case func_kind::select:
            return "select";
        case func_kind::logical_and:
            return "logical_and";
        case func_kind::logical_or:
            return "logical_or";
        case func_kind::rel_eq:
            return "rel_eq";
        case func_kind::rel_neq:
            return "rel_neq";
        case func_kind::rel_lt:
            return "rel_lt";
        case func_kind::rel_gt:
            return "rel_gt";
        case func_kind::rel_lte:
            return "rel_lte";
        case func_kind::rel_gte:
            return "rel_gte";
        case func_kind::asinh:
            return "asinh";
        case func_kind::acosh:
            return "acosh";
        case func_kind::atanh:
            return "atanh";
        case func_kind::erf:
            return "erf";
        case func_kind::sigmoid:
            return "sigmoid";
        case func_kind::custom:
            return "custom";
    }
    return "?";
}

namespace
{

inline void hash_combine(std::size_t &seed, std::size_t v)
{
    seed ^= v + 0x9e3779b97f4a7c15ull + (seed << 6) + (seed >> 2);
}

std::size_t hash_double(double x)
{
    // +0 and -0 compare equal: hash them identically.
    if (x == 0) {
        x = 0;
    }
    std::uint64_t bits = 0;
    std::memcpy(&bits, &x, sizeof(bits));
    return std::hash<std::uint64_t>{}(bits);
}

} // namespace

func::func(func_kind k, std::vector<expression> args, std::uint32_t rule)
{
    auto node = std::make_shared<func_node>();
    node->kind = k;
    node->args = std::move(args);
    node->rule = rule;
    std::size_t h = std::hash<int>{}(static_cast<int>(k) + 17 + static_cast<int>(rule) * 64);
    for (const auto &a : node->args) {
        hash_combine(h, a.hash());
    }
    node->hash = h;
    m_ptr = std::move(node);
}

func func::copy_with_new_args(std::vector<expression> new_args) const
{
    return func(kind(), std::move(new_args), rule());
}

std::string func::name() const
{
    return kind() == func_kind::custom ? get_node_rule(rule()).name : std::string(func_kind_name(kind()));
}

std::size_t expression::hash() const
{
    switch (m_value.index()) {
        case 0u:
            return hash_double(num());
        case 1u:
            return std::hash<std::string>{}(var_name()) ^ 0x51ed270b;
        case 2u:
            return std::hash<std::uint32_t>{}(par_idx()) ^ 0x2545f491;
        default:
            return fn().hash();
    }
}

bool operator==(const expression &a, const expression &b)
{
    if (a.value().index() != b.value().index()) {
        return false;
    }
    switch (a.value().index()) {
        case 0u: {
            // NOTE: nan numbers compare equal to each other, so that CSE is well-behaved
            // (reference: number equality in src/number.cpp).
            const auto x = a.num(), y = b.num();
            return x == y || (std::isnan(x) && std::isnan(y));
        }
        case 1u:
            return a.var_name() == b.var_name();
        case 2u:
            return a.par_idx() == b.par_idx();
        default: {
            const auto &fa = a.fn();
            const auto &fb = b.fn();
            if (fa.get_ptr() == fb.get_ptr()) {
                return true;
            }
            if (fa.hash() != fb.hash() || fa.kind() != fb.kind() || fa.rule() != fb.rule()
                || fa.args().size() != fb.args().size()) {
                return false;
            }
            for (std::size_t i = 0; i < fa.args().size(); ++i) {
                if (!(fa.args()[i] == fb.args()[i])) {
                    return false;
                }
            }
            return true;
        }
    }
}

std::string expression::to_string() const
{
    std::ostringstream oss;
    switch (m_value.index()) {
        case 0u:
            oss << std::setprecision(17) << num();
            break;
        case 1u:
            oss << var_name();
            break;
        case 2u:
            oss << "p" << par_idx();
            break;
        default: {
            const auto &f = fn();
            oss << f.name() << '(';
            for (std::size_t i = 0; i < f.args().size(); ++i) {
                if (i != 0u) {
                    oss << ", ";
                }
                oss << f.args()[i].to_string();
            }
            oss << ')';
        }
    }
    return oss.str();
}

namespace detail
{

expression make_func(func_kind k, std::vector<expression> args)
{
    return expression{func(k, std::move(args))};
}

expression sub(expression a, expression b)
{
    return make_func(func_kind::sub, {std::move(a), std::move(b)});
}

expression div(expression a, expression b)
{
    return make_func(func_kind::div, {std::move(a), std::move(b)});
}

expression sum_sq(std::vector<expression> args)
{
    return make_func(func_kind::sum_sq, std::move(args));
}

expression num_identity(expression e)
{
    return make_func(func_kind::num_identity, {std::move(e)});
}

} // namespace detail

const expression time = detail::make_func(func_kind::time, {});

// --- Operators (reference: src/expression_ops.cpp:34-91). ---
expression operator+(expression e)
{
    return e;
}

expression operator-(const expression &e)
{
    if (e.is_number()) {
        return expression{-e.num()};
    }
    return prod({expression{-1.}, e});
}

expression operator+(const expression &a, const expression &b)
{
    if (a.is_number() && b.is_number()) {
        return expression{a.num() + b.num()};
    }
    return sum({a, b});
}

expression operator-(const expression &a, const expression &b)
{
    if (a.is_number() && b.is_number()) {
        return expression{a.num() - b.num()};
    }
    return a + -b;
}

expression operator*(const expression &a, const expression &b)
{
    if (a.is_number() && b.is_number()) {
        return expression{a.num() * b.num()};
    }
    return prod({a, b});
}

expression operator/(const expression &a, const expression &b)
{
    if (a.is_number() && b.is_number()) {
        return expression{a.num() / b.num()};
    }
    return prod({a, pow(b, expression{-1.})});
}

namespace
{

// sum() and prod() share their canonical form (src/math/sum.cpp:548-601, src/math/prod.cpp:913-973): the numerical
// arguments are combined, in their order of appearance, into ONE number which leads the remaining arguments (these keep
// their order); a number equal to the neutral element is dropped, and for products a zero swallows everything. No
// arguments left: the neutral element; one: that argument itself, without a function around it.
expression commutative(func_kind kind, const std::vector<expression> &args, double neutral, bool zero_absorbs,
                       double (*combine)(double, double))
{
    std::vector<expression> rest;
    rest.reserve(args.size() + 1u);
    bool any_number = false;
    double folded = neutral;
    for (const auto &a : args) {
        if (a.is_number()) {
            folded = any_number ? combine(folded, a.num()) : a.num();
            any_number = true;
        } else {
            rest.push_back(a);
        }
    }
    if (any_number && zero_absorbs && folded == 0) {
        return expression{folded};
    }
    const bool keep_number = any_number && !(folded == neutral);
    if (rest.empty()) {
        return expression{any_number ? folded : neutral};
    }
    if (!keep_number && rest.size() == 1u) {
        return rest[0];
    }
    if (keep_number) {
        rest.insert(rest.begin(), expression{folded});
    }
    return detail::make_func(kind, std::move(rest));
}

} // namespace

expression sum(std::vector<expression> args)
{
    return commutative(func_kind::sum, args, 0., false, [](double x, double y) { return x + y; });
}

expression prod(std::vector<expression> args)
{
    return commutative(func_kind::prod, args, 1., true, [](double x, double y) { return x * y; });
}

// Reference: src/math/pow.cpp:1024-1062.
expression pow(const expression &b, const expression &e)
{
    if (b.is_number() && e.is_number()) {
        return expression{std::pow(b.num(), e.num())};
    }
    if (e.is_number()) {
        if (e.num() == 0) {
            return expression{1.};
        }
        if (e.num() == 1) {
            return b;
        }
    }
    return detail::make_func(func_kind::pow, {b, e});
}

// Reference: src/math/sqrt.cpp:16-19.
expression sqrt(const expression &e)
{
    return pow(e, expression{.5});
}

expression square(const expression &e)
{
    return pow(e, expression{2.});
}

// Reference: src/math/sin.cpp:381-395, src/math/cos.cpp:381-395.
expression sin(expression e)
{
    if (e.is_number()) {
        return expression{std::sin(e.num())};
    }
    return detail::make_func(func_kind::sin, {std::move(e)});
}

expression cos(expression e)
{
    if (e.is_number()) {
        return expression{std::cos(e.num())};
    }
    return detail::make_func(func_kind::cos, {std::move(e)});
}

expression exp(expression e)
{
    if (e.is_number()) {
        return expression{std::exp(e.num())};
    }
    return detail::make_func(func_kind::exp, {std::move(e)});
}

expression log(expression e)
{
    if (e.is_number()) {
        return expression{std::log(e.num())};
    }
    return detail::make_func(func_kind::log, {std::move(e)});
}

// Unary elementary functions: constant folding on numbers, otherwise a function node
// (reference: src/math/tan.cpp etc., e.g. the number overloads at src/math/tan.cpp:230-245).
#define HEYOKA_AMD_UNARY_FUNC(name, eval)                                                                              \
    expression name(expression e)                                                                                      \
    {                                                                                                                  \
        if (e.is_number()) {                                                                                           \
            const double x = e.num();                                                                                  \
            return expression{eval};                                                                                   \
        }                                                                                                              \
        return detail::make_func(func_kind::name, {std::move(e)});                                                     \
    }

HEYOKA_AMD_UNARY_FUNC(tan, std::tan(x))
HEYOKA_AMD_UNARY_FUNC(tanh, std::tanh(x))
HEYOKA_AMD_UNARY_FUNC(sinh, std::sinh(x))
HEYOKA_AMD_UNARY_FUNC(cosh, std::cosh(x))
HEYOKA_AMD_UNARY_FUNC(asin, std::asin(x))
HEYOKA_AMD_UNARY_FUNC(acos, std::acos(x))
HEYOKA_AMD_UNARY_FUNC(atan, std::atan(x))
HEYOKA_AMD_UNARY_FUNC(asinh, std::asinh(x))
HEYOKA_AMD_UNARY_FUNC(acosh, std::acosh(x))
HEYOKA_AMD_UNARY_FUNC(atanh, std::atanh(x))
HEYOKA_AMD_UNARY_FUNC(erf, std::erf(x))
HEYOKA_AMD_UNARY_FUNC(sigmoid, 1. / (1. + std::exp(-x)))

#undef HEYOKA_AMD_UNARY_FUNC

// Reference: src/math/atan2.cpp:763-786 (two numbers fold).
expression atan2(expression y, expression x)
{
    if (y.is_number() && x.is_number()) {
        return expression{std::atan2(y.num(), x.num())};
    }
    return detail::make_func(func_kind::atan2, {std::move(y), std::move(x)});
}

// Reference: src/math/kepE.cpp:801-809 (zero eccentricity: E = M; no other folding).
expression kepE(expression e, expression M)
{
    if (e.is_number() && e.num() == 0) {
        return M;
    }
    return detail::make_func(func_kind::kepE, {std::move(e), std::move(M)});
}

namespace
{

// Reference: relu_slope_check(), src/math/relu.cpp:52-59.
void relu_slope_check(double slope)
{
    if (!std::isfinite(slope) || slope < 0) {
        std::ostringstream oss;
        oss << "The slope parameter for a leaky ReLU must be finite and non-negative, but the value " << slope
            << " was provided instead";
        throw std::invalid_argument(oss.str());
    }
}

} // namespace

// Reference: src/math/relu.cpp:580-602 (numbers fold).
expression relu(expression x, double slope)
{
    relu_slope_check(slope);
    if (x.is_number()) {
        return expression{x.num() > 0 ? x.num() : slope * x.num()};
    }
    return detail::make_func(func_kind::relu, {std::move(x), expression{slope}});
}

expression relup(expression x, double slope)
{
    relu_slope_check(slope);
    if (x.is_number()) {
        return expression{x.num() > 0 ? 1. : slope};
    }
    return detail::make_func(func_kind::relup, {std::move(x), expression{slope}});
}

leaky_relu::leaky_relu(double s) : slope(s)
{
    relu_slope_check(s);
}

leaky_relup::leaky_relup(double s) : slope(s)
{
    relu_slope_check(s);
}

// Reference: src/math/select.cpp:267-270 (no folding).
expression select(expression cond, expression t, expression f)
{
    return detail::make_func(func_kind::select, {std::move(cond), std::move(t), std::move(f)});
}

// Reference: src/math/logical.cpp:314-338.
expression logical_and(std::vector<expression> args)
{
    if (args.empty()) {
        return expression{1.};
    }
    if (args.size() == 1u) {
        return std::move(args[0]);
    }
    return detail::make_func(func_kind::logical_and, std::move(args));
}

expression logical_or(std::vector<expression> args)
{
    if (args.empty()) {
        return expression{0.};
    }
    if (args.size() == 1u) {
        return std::move(args[0]);
    }
    return detail::make_func(func_kind::logical_or, std::move(args));
}

// Reference: src/math/relational.cpp:343-354 (no folding).
#define HEYOKA_AMD_REL_FUNC(name)                                                                                      \
    expression name(expression a, expression b)                                                                        \
    {                                                                                                                  \
        return detail::make_func(func_kind::rel_##name, {std::move(a), std::move(b)});                                 \
    }
HEYOKA_AMD_REL_FUNC(eq)
HEYOKA_AMD_REL_FUNC(neq)
HEYOKA_AMD_REL_FUNC(lt)
HEYOKA_AMD_REL_FUNC(gt)
HEYOKA_AMD_REL_FUNC(lte)
HEYOKA_AMD_REL_FUNC(gte)
#undef HEYOKA_AMD_REL_FUNC

// --- Traversal. ---
// The distinct function nodes below root which are not done yet, every node behind its arguments; the arguments of a node
// are taken from the LAST to the first (this is the order in which the reference numbers the u variables of a
// decomposition, src/expression_decompose.cpp:43-210, and visits the nodes in its transformations,
// src/detail/ex_traversal.cpp:35-180). A stack of (node, number of arguments already descended into).
std::vector<const expression *> function_nodes_postorder(const expression &root,
                                                         const std::function<bool(const void *)> &is_done)
{
    std::vector<const expression *> order;
    if (!root.is_func() || is_done(root.fn().get_ptr())) {
        return order;
    }
    std::set<const void *> finished;
    struct frame {
        const expression *node;
        std::size_t descended;
    };
    std::vector<frame> path{{&root, 0u}};
    while (!path.empty()) {
        auto &top = path.back();
        const auto &args = top.node->fn().args();
        if (top.descended == args.size()) {
            finished.insert(top.node->fn().get_ptr());
            order.push_back(top.node);
            path.pop_back();
            continue;
        }
        const auto &next = args[args.size() - 1u - top.descended];
        ++top.descended;
        if (next.is_func()) {
            const auto *id = next.fn().get_ptr();
            if (finished.count(id) == 0u && !is_done(id)) {
                path.push_back({&next, 0u});
            }
        }
    }
    return order;
}

// Bottom-up rebuild: the leaves through leaf_tfunc, every function node (with its rebuilt arguments) through
// branch_tfunc; a node whose arguments did not change identity is not copied. cache: node identity -> result, shared by
// the calls which belong together.
expression traverse_transform_nodes(ptr_ex_map &cache, const expression &e,
                                    const std::function<expression(const expression &)> &leaf_tfunc,
                                    const std::function<expression(const expression &)> &branch_tfunc)
{
    if (!e.is_func()) {
        return leaf_tfunc ? leaf_tfunc(e) : e;
    }
    const auto rebuilt = [&](const expression &a) -> expression {
        if (a.is_func()) {
            return cache.at(a.fn().get_ptr());
        }
        return leaf_tfunc ? leaf_tfunc(a) : a;
    };
    for (const auto *node : function_nodes_postorder(e, [&cache](const void *id) { return cache.count(id) != 0u; })) {
        const auto &f = node->fn();
        std::vector<expression> args;
        args.reserve(f.args().size());
        bool untouched = true;
        for (const auto &a : f.args()) {
            args.push_back(rebuilt(a));
            const auto &b = args.back();
            if (a.is_func() != b.is_func()) {
                untouched = false;
            } else if (a.is_func()) {
                untouched = untouched && a.fn().get_ptr() == b.fn().get_ptr();
            } else {
                untouched = untouched && a == b;
            }
        }
        auto result = untouched ? *node : expression{f.copy_with_new_args(std::move(args))};
        if (branch_tfunc) {
            result = branch_tfunc(result);
        }
        cache.emplace(f.get_ptr(), std::move(result));
    }
    return cache.at(e.fn().get_ptr());
}

namespace
{

void visit_leaves(std::set<const void *> &seen, const expression &e, const std::function<void(const expression &)> &vf)
{
    std::vector<const expression *> stack{&e};
    while (!stack.empty()) {
        const auto *cur = stack.back();
        stack.pop_back();
        if (cur->is_func()) {
            const auto &f = cur->fn();
            if (!seen.insert(f.get_ptr()).second) {
                continue;
            }
            for (const auto &a : f.args()) {
                stack.push_back(&a);
            }
        } else {
            vf(*cur);
        }
    }
}

} // namespace

std::vector<std::string> get_variables(const std::vector<expression> &v_ex)
{
    std::set<std::string> s;
    std::set<const void *> seen;
    for (const auto &e : v_ex) {
        visit_leaves(seen, e, [&s](const expression &l) {
            if (l.is_variable()) {
                s.insert(l.var_name());
            }
        });
    }
    return {s.begin(), s.end()};
}

std::vector<std::string> get_variables(const expression &e)
{
    return get_variables(std::vector<expression>{e});
}

expression rename_variables(ptr_ex_map &cache, const expression &e,
                            const std::unordered_map<std::string, std::string> &repl)
{
    return traverse_transform_nodes(
        cache, e,
        [&repl](const expression &l) {
            if (l.is_variable()) {
                if (const auto it = repl.find(l.var_name()); it != repl.end()) {
                    return expression{it->second};
                }
            }
            return l;
        },
        {});
}

std::vector<expression> rename_variables(const std::vector<expression> &v_ex,
                                         const std::unordered_map<std::string, std::string> &repl)
{
    ptr_ex_map cache;
    std::vector<expression> ret;
    ret.reserve(v_ex.size());
    for (const auto &e : v_ex) {
        ret.push_back(rename_variables(cache, e, repl));
    }
    return ret;
}

std::uint32_t get_param_size(const std::vector<expression> &v_ex)
{
    std::uint32_t ret = 0;
    std::set<const void *> seen;
    for (const auto &e : v_ex) {
        visit_leaves(seen, e, [&ret](const expression &l) {
            if (l.is_param()) {
                if (l.par_idx() == UINT32_MAX) {
                    throw std::overflow_error("Overflow detected in get_param_size()");
                }
                ret = std::max(ret, l.par_idx() + 1u);
            }
        });
    }
    return ret;
}

bool is_time_dependent(const std::vector<expression> &v_ex)
{
    std::set<const void *> seen;
    std::vector<const expression *> stack;
    for (const auto &e : v_ex) {
        stack.push_back(&e);
    }
    while (!stack.empty()) {
        const auto *cur = stack.back();
        stack.pop_back();
        if (cur->is_func()) {
            const auto &f = cur->fn();
            if (f.kind() == func_kind::time) {
                return true;
            }
            if (!seen.insert(f.get_ptr()).second) {
                continue;
            }
            for (const auto &a : f.args()) {
                stack.push_back(&a);
            }
        }
    }
    return false;
}

} // namespace heyoka_amd
