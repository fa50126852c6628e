// Registry of elementary-function rules: the per-node extension seam of the integrator.
//
// The reference lets any func_base-derived class supply its own Taylor decomposition (hidden dependencies), its
// order-k derivative rule and the function which the compact mode calls for it (include/heyoka/func.hpp:94-96,
// :117-147; default decomposition src/func.cpp:392-420; a function without a rule raises not_implemented_error,
// func.hpp:266-267). Here the counterpart of a func_base subclass is a node_rule:
//   * name / number of arguments;
//   * decompose(): the definitions of the hidden u variables which are appended right behind the node itself (each ONE
//     elementary function of the node, its arguments and the hidden variables defined before it), the dependencies
//     between them, and the hidden dependencies of the node in the order its rule reads them - the information a
//     func_base::taylor_decompose() override encodes by appending to the decomposition by hand;
//   * hip_source: HIP device code with the two functions every generator calls,
//         double hy_rule_<name>_order0(const double *x);
//             x[i] = order-0 value of argument i (a number / parameter argument is passed by value as well); the
//             generators call it through an out-of-line frame (hy_rule_<name>_value(x0, x1, ...), generated): library
//             calls and data-dependent loops are welcome here and stay out of the stepper's own control flow;
//         double hy_rule_<name>_orderk(unsigned k, const hy_jet &a, const hy_jet *x, const hy_jet *h);
//             k >= 1; a = the node's own coefficients (orders 0 .. k - 1), x[i] = coefficients of argument i (orders
//             0 .. k; a number or parameter has order 0 only), h[j] = coefficients of hidden dependency j (orders
//             0 .. k - 1). hy_jc(jet, j) reads coefficient j (0 beyond the last valid order).
//     The same source serves the straight-line generator (the jets are local arrays of SSA values: after inlining the
//     loops are unrolled and the arrays dissolve into registers), the interpreted (table) steppers in both variants (the
//     jets are strided views of the tape), the dense event-jet kernel and compiled functions (order 0 only).
// Rules are registered once per process (register_node_rule(); C ABI hy_node_rule_register(), include/heyoka_amd.h) and
// referred to by name: custom_func("kepF", {h, k, lam}). Unknown names raise not_implemented_error when the expression is
// built - like the reference, never later. kepF and kepDE are defined this way, with nothing but this interface
// (builtin_rules.cpp).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "expression.hpp"

namespace heyoka_amd
{

struct hidden_def {
    // One elementary function of `self`, the arguments and hidden(j), j < own index.
    expression ex;
    // Hidden dependencies of THIS definition (indices into the list of hidden definitions), in the order its own rule
    // reads them: e.g. sin(self) depends on the entry holding cos(self) and vice versa.
    std::vector<std::uint32_t> deps;
};

struct node_rule {
    std::string name;
    std::uint32_t n_args = 0;
    // self: the u variable of the node; args: its (already decomposed) arguments - variables, numbers or parameters;
    // hidden(j): the u variable of the j-th hidden definition.
    std::function<std::vector<hidden_def>(const expression &self, const std::vector<expression> &args,
                                          const std::function<expression(std::uint32_t)> &hidden)>
        decompose;
    // Hidden dependencies of the node (indices into the list returned by decompose()), in the order hy_rule_*_orderk reads
    // them as h[0], h[1], ...
    std::vector<std::uint32_t> deps;
    std::string hip_source;
    // Optional simplification at construction (e.g. kepDE(0, 0, DM) = DM): returns true and sets `out` to replace the
    // function by another expression.
    std::function<bool(const std::vector<expression> &args, expression &out)> fold;
};

// Returns the id of the rule (>= 1). Throws std::invalid_argument on an invalid rule or a duplicate name.
std::uint32_t register_node_rule(node_rule);
// nullptr if there is no such rule.
const node_rule *find_node_rule(const std::string &name);
const node_rule &get_node_rule(std::uint32_t id);
std::uint32_t node_rule_id(const std::string &name); // 0 if unknown

// f(args) for a registered rule; not_implemented_error for an unknown name (reference: func.hpp:266-267).
expression custom_func(const std::string &name, std::vector<expression> args);

// The two functions defined through the registry alone (builtin_rules.cpp): the eccentric longitude F(h, k, lam),
// F + h cos F - k sin F = lam (reference: src/math/kepF.cpp), and the eccentric-anomaly difference DE(s0, c0, DM),
// DE - c0 sin DE + s0 (1 - cos DE) = DM (reference: src/math/kepDE.cpp - there without a Taylor rule).
// (Registers kepF and kepDE on first use.)
void ensure_builtin_rules();
expression kepF(expression h, expression k, expression lam);
expression kepDE(expression s0, expression c0, expression DM);
// The constant pi as a function without arguments which occupies its own u variable, like the reference's
// heyoka::pi (include/heyoka/math/constants.hpp:48-60, :117; src/math/constants.cpp:258-273: order 0 = the value, 0
// beyond) - the third function defined through the registry alone.
expression pi_constant();

// The device-side interface of the rules (struct hy_jet, hy_jc) + the sources of the rules used by a program, for the
// generators: emitted once per module, behind the prelude.
std::string node_rules_device_source(const std::vector<std::uint32_t> &rule_ids);

} // namespace heyoka_amd
