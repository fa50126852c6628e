// Taylor decomposition of an ODE system into elementary subexpressions ("u variables").
//
// Reference semantics: taylor_decompose_sys() and its rewrite pipeline,
// src/taylor_01.cpp:848-1008 (+ CSE :315-443, BFS topological sort :454-645),
// src/math/sum.cpp:428-544, src/math/prod.cpp:736-908, src/expression_basic.cpp:1177-1231,
// include/heyoka/detail/udf_split.hpp:49-100, src/expression_decompose.cpp:43-210.
#pragma once

#include <cstdint>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "expression.hpp"

namespace heyoka_amd
{

// Same shape as the reference's taylor_dc_t (include/heyoka/detail/fwd_decl.hpp:63):
// first n_eq entries = state variables, middle = one elementary function each (+ hidden deps),
// last n_eq entries = definitions of the derivatives of the state variables.
using taylor_dc_t = std::vector<std::pair<expression, std::vector<std::uint32_t>>>;

taylor_dc_t taylor_decompose_sys(const std::vector<std::pair<expression, expression>> &sys);
// With extra functions of the state (e.g. event equations): sv_funcs_dc receives the index of the u variable holding
// each of them (reference: src/taylor_01.cpp:848-1008, second return value).
taylor_dc_t taylor_decompose_sys(const std::vector<std::pair<expression, expression>> &sys,
                                 const std::vector<expression> &sv_funcs, std::vector<std::uint32_t> &sv_funcs_dc);

// Decomposition of a vector function of the variables `vars` (reference: function_decompose(),
// src/expression_cfunc.cpp:723-900): vars.size() leading entries, fn.size() trailing definitions.
taylor_dc_t function_decompose(const std::vector<expression> &fn, const std::vector<expression> &vars);

// Reference: detail::validate_ode_sys() (src/detail/validate_ode_sys.cpp).
void validate_ode_sys(const std::vector<std::pair<expression, expression>> &sys);

// Parse the index out of a "u_123" variable name.
std::uint32_t uname_to_index(const std::string &);

// --- Flattened, backend-facing form of a decomposition. ---
struct operand {
    enum class kind : std::uint8_t { uvar = 0, num = 1, par = 2 };
    kind type = kind::num;
    std::uint32_t idx = 0; // u-variable index or parameter index.
    double value = 0;      // numerical value (type == num).
};

struct dc_node {
    func_kind kind;
    std::uint32_t rule = 0; // id of the node rule (kind == custom)
    std::vector<operand> args;
    std::vector<std::uint32_t> deps; // hidden dependencies (u-variable indices).
};

struct taylor_program {
    std::uint32_t n_eq = 0;
    std::uint32_t n_u = 0;
    std::uint32_t n_par = 0;
    bool time_dependent = false;
    // nodes[i] defines u variable n_eq + i.
    std::vector<dc_node> nodes;
    // sv_defs[i] = definition of the time derivative of state variable i.
    std::vector<operand> sv_defs;
    // u variables holding the event equations (terminal events first), empty if no events.
    std::vector<std::uint32_t> ev_u;
};

// NOTE: n_outs = number of trailing definitions (defaults to n_eq, i.e. a Taylor decomposition).
taylor_program make_program(const taylor_dc_t &dc, std::uint32_t n_eq,
                            std::uint32_t n_outs = std::numeric_limits<std::uint32_t>::max());

// u variables which do not depend on the state or on time: every argument is a number, a parameter or another constant
// u variable (e.g. -par[0], par[0] + par[1] in models with runtime masses). All their Taylor coefficients beyond order 0
// vanish: a product with such a factor is linear in the other one.
inline std::vector<char> constant_uvars(const taylor_program &p)
{
    std::vector<char> c(p.n_u, 0);
    for (std::size_t i = 0; i < p.nodes.size(); ++i) {
        const auto &n = p.nodes[i];
        if (n.kind == func_kind::time || !n.deps.empty()) {
            continue;
        }
        bool all_const = true;
        for (const auto &o : n.args) {
            if (o.type == operand::kind::uvar && (o.idx < p.n_eq || c[o.idx] == 0)) {
                all_const = false;
            }
        }
        c[p.n_eq + i] = all_const ? 1 : 0;
    }
    return c;
}

// Order of the Taylor method from the tolerance (reference: include/heyoka/detail/taylor_common.hpp:165-191).
std::uint32_t taylor_order_from_tol(double tol);

} // namespace heyoka_amd
