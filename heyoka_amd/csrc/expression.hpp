// Host-side symbolic expression system for the MI355X batch Taylor integrator.
//
// Mirrors the *semantics* of heyoka's expression layer (reference:
// include/heyoka/expression.hpp:73-118, src/expression_ops.cpp:45-91,
// src/math/prod.cpp:913-973, src/math/sum.cpp:548-601, src/math/pow.cpp:1024-1062)
// for the node types reachable from the hot path of taylor_adaptive_batch<double>.
// This is a from-scratch implementation: a single tagged node type with shared,
// immutable function nodes (identity = pointer, equality = structure).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <variant>
#include <vector>

namespace heyoka_amd
{

struct not_implemented_error final : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Kinds of elementary functions known to the Taylor decomposition.
enum class func_kind : std::uint8_t {
    sum,
    prod,
    pow,
    sub,
    div,
    sum_sq,
    sin,
    cos,
    exp,
    log,
    time,
    num_identity,
    // Elementary functions beyond the N-body set (reference: src/math/{tan,tanh,sinh,cosh,asin,acos,atan,
    // asinh,acosh,atanh,erf,sigmoid}.cpp).
    tan,
    tanh,
    sinh,
    cosh,
    asin,
    acos,
    atan,
    asinh,
    acosh,
    atanh,
    erf,
    sigmoid,
    // Two-argument functions (reference: src/math/atan2.cpp, src/math/kepE.cpp).
    atan2,
    kepE,
    // Piecewise functions (reference: src/math/{relu,select,relational,logical}.cpp). relu / relup carry the slope
    // of the leaky variants as a second (numerical) argument.
    relu,
    relup,
    select,
    logical_and,
    logical_or,
    rel_eq,
    rel_neq,
    rel_lt,
    rel_gt,
    rel_lte,
    rel_gte,
    // A function defined through the registry of node rules (node_rule.hpp): func::rule() identifies it.
    custom
};

const char *func_kind_name(func_kind);

class expression;

struct number {
    double value = 0;
};

struct variable {
    std::string name;
};

struct param {
    std::uint32_t idx = 0;
};

// Immutable function node. Shared between all expressions referring to it.
struct func_node {
    func_kind kind;
    std::vector<expression> args;
    std::size_t hash = 0;
    std::uint32_t rule = 0; // id of the node rule (kind == custom), 0 otherwise
};

class func
{
    std::shared_ptr<const func_node> m_ptr;

public:
    func(func_kind, std::vector<expression>, std::uint32_t rule = 0);

    [[nodiscard]] func_kind kind() const
    {
        return m_ptr->kind;
    }
    [[nodiscard]] const std::vector<expression> &args() const
    {
        return m_ptr->args;
    }
    [[nodiscard]] std::uint32_t rule() const
    {
        return m_ptr->rule;
    }
    // Name of the function (func_kind_name() for the built-in kinds, the rule's name for a custom function).
    [[nodiscard]] std::string name() const;
    // Identity of the node (used by the traversal caches, reference: func::get_ptr()).
    [[nodiscard]] const void *get_ptr() const
    {
        return m_ptr.get();
    }
    [[nodiscard]] std::size_t hash() const
    {
        return m_ptr->hash;
    }
    [[nodiscard]] func copy_with_new_args(std::vector<expression>) const;
};

class expression
{
public:
    using value_type = std::variant<number, variable, param, func>;

private:
    value_type m_value;

public:
    expression() : m_value(number{0.}) {}
    expression(double x) : m_value(number{x}) {} // NOLINT
    explicit expression(number n) : m_value(n) {}
    explicit expression(variable v) : m_value(std::move(v)) {}
    explicit expression(std::string name) : m_value(variable{std::move(name)}) {}
    explicit expression(const char *name) : m_value(variable{name}) {}
    explicit expression(param p) : m_value(p) {}
    explicit expression(func f) : m_value(std::move(f)) {}

    [[nodiscard]] const value_type &value() const
    {
        return m_value;
    }

    [[nodiscard]] bool is_number() const
    {
        return m_value.index() == 0u;
    }
    [[nodiscard]] bool is_variable() const
    {
        return m_value.index() == 1u;
    }
    [[nodiscard]] bool is_param() const
    {
        return m_value.index() == 2u;
    }
    [[nodiscard]] bool is_func() const
    {
        return m_value.index() == 3u;
    }
    [[nodiscard]] double num() const
    {
        return std::get<number>(m_value).value;
    }
    [[nodiscard]] const std::string &var_name() const
    {
        return std::get<variable>(m_value).name;
    }
    [[nodiscard]] std::uint32_t par_idx() const
    {
        return std::get<param>(m_value).idx;
    }
    [[nodiscard]] const func &fn() const
    {
        return std::get<func>(m_value);
    }

    [[nodiscard]] std::size_t hash() const;
    [[nodiscard]] std::string to_string() const;
};

bool operator==(const expression &, const expression &);
inline bool operator!=(const expression &a, const expression &b)
{
    return !(a == b);
}

struct expression_hash {
    std::size_t operator()(const expression &e) const
    {
        return e.hash();
    }
};

// --- Arithmetic with the reference's constant folding / canonicalisation. ---
expression operator+(expression);
expression operator-(const expression &);
expression operator+(const expression &, const expression &);
expression operator-(const expression &, const expression &);
expression operator*(const expression &, const expression &);
expression operator/(const expression &, const expression &);
// NOTE: no double overloads needed, expression is implicitly constructible from double.

expression sum(std::vector<expression>);
expression prod(std::vector<expression>);
expression pow(const expression &, const expression &);
expression sqrt(const expression &);
expression square(const expression &);
expression sin(expression);
expression cos(expression);
expression exp(expression);
expression log(expression);
expression tan(expression);
expression tanh(expression);
expression sinh(expression);
expression cosh(expression);
expression asin(expression);
expression acos(expression);
expression atan(expression);
expression asinh(expression);
expression acosh(expression);
expression atanh(expression);
expression erf(expression);
expression sigmoid(expression);
// atan2(y, x) (src/math/atan2.cpp:763-786) and the eccentric anomaly E(e, M), E - e sin E = M
// (src/math/kepE.cpp:801-809).
expression atan2(expression y, expression x);
expression kepE(expression e, expression M);
// relu(x, slope) = x > 0 ? x : slope * x and its derivative relup (src/math/relu.cpp:580-602); leaky_relu(slope)(x).
expression relu(expression x, double slope = 0.);
expression relup(expression x, double slope = 0.);
struct leaky_relu {
    double slope;
    explicit leaky_relu(double s);
    expression operator()(expression x) const
    {
        return relu(std::move(x), slope);
    }
};
struct leaky_relup {
    double slope;
    explicit leaky_relup(double s);
    expression operator()(expression x) const
    {
        return relup(std::move(x), slope);
    }
};
// select(c, t, f) = c != 0 ? t : f (src/math/select.cpp:267-270), logical_and / logical_or of the truth values
// (src/math/logical.cpp:314-338), comparisons returning 1 / 0 (src/math/relational.cpp:343-354).
expression select(expression cond, expression t, expression f);
expression logical_and(std::vector<expression> args);
expression logical_or(std::vector<expression> args);
expression eq(expression, expression);
expression neq(expression, expression);
expression lt(expression, expression);
expression gt(expression, expression);
expression lte(expression, expression);
expression gte(expression, expression);

namespace detail
{
// Non-folding constructors used by the decomposition rewrites.
expression sub(expression, expression);
expression div(expression, expression);
expression sum_sq(std::vector<expression>);
expression num_identity(expression);
expression make_func(func_kind, std::vector<expression>);
} // namespace detail

// The time placeholder (reference: include/heyoka/math/time.hpp, `heyoka::time`).
extern const expression time;

// Runtime parameters: par[i] (reference: include/heyoka/param.hpp).
struct par_impl {
    expression operator[](std::uint32_t i) const
    {
        return expression{param{i}};
    }
};
inline constexpr par_impl par{};

// "x"_var, 2_dbl / 1.5_dbl (reference: inline namespace literals, include/heyoka/expression.hpp:122-146).
inline namespace literals
{

inline expression operator""_var(const char *s, std::size_t n)
{
    return expression{std::string(s, n)};
}
inline expression operator""_dbl(long double x)
{
    return expression{static_cast<double>(x)};
}
inline expression operator""_dbl(unsigned long long n)
{
    return expression{static_cast<double>(n)};
}

} // namespace literals

// make_vars("x", "v") -> array of variable expressions; make_vars("x") -> a single expression
// (include/heyoka/expression.hpp:540-549).
template <typename Arg0, typename... Args>
inline auto make_vars(const Arg0 &name, const Args &...names)
{
    if constexpr (sizeof...(Args) == 0u) {
        return expression{std::string(name)};
    } else {
        return std::array<expression, sizeof...(Args) + 1u>{expression{std::string(name)}, expression{std::string(names)}...};
    }
}

// prime(x) = rhs -> pair(x, rhs).
struct prime_wrapper {
    expression lhs;
    std::pair<expression, expression> operator=(expression rhs) &&
    {
        return {std::move(lhs), std::move(rhs)};
    }
};
inline prime_wrapper prime(expression e)
{
    if (!e.is_variable()) {
        throw std::invalid_argument("Cannot apply the prime() operator to a non-variable expression");
    }
    return prime_wrapper{std::move(e)};
}

// --- Traversal helpers. ---
using ptr_ex_map = std::unordered_map<const void *, expression>;

// The distinct function nodes below root for which is_done() is false, arguments before the functions which use them, the
// arguments of a node from the last to the first (the numbering order of the reference's decomposition).
std::vector<const expression *> function_nodes_postorder(const expression &root,
                                                         const std::function<bool(const void *)> &is_done);

// Post-order transform of the function nodes of e (children first, in the reference's visiting order:
// last argument first). Shared nodes are transformed once (reference: src/detail/ex_traversal.cpp:35-180).
expression traverse_transform_nodes(ptr_ex_map &cache, const expression &e,
                                    const std::function<expression(const expression &)> &leaf_tfunc,
                                    const std::function<expression(const expression &)> &branch_tfunc);

std::vector<std::string> get_variables(const expression &);
std::vector<std::string> get_variables(const std::vector<expression> &);
expression rename_variables(ptr_ex_map &cache, const expression &,
                            const std::unordered_map<std::string, std::string> &);
std::vector<expression> rename_variables(const std::vector<expression> &,
                                         const std::unordered_map<std::string, std::string> &);
// Number of runtime parameters = 1 + max param index (0 if none).
std::uint32_t get_param_size(const std::vector<expression> &);
bool is_time_dependent(const std::vector<expression> &);

} // namespace heyoka_amd
