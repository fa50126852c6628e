// "Cluster" code generation mode: one ODE system per group of L lanes, jets resident in registers.
//
// The Taylor recursion of a nonlinear node (product, sum of squares, pow, sin/cos, ...) at order k
// reads *all* lower orders of its operands: for an N-body system that history (15 pairs x 6 jets x
// 20 orders for the outer Solar System) cannot live in the registers of a single lane, and parking
// it in HBM makes the step bandwidth-bound at a small fraction of the FP64 rate. This generator
// instead partitions the decomposition (reference: taylor_dc_t, src/taylor_01.cpp:848-1008) into
//
//   * clusters - connected components under "history" edges (a nonlinear node and the operands whose
//     full history it needs). All clusters must be isomorphic (one code path, SIMT-friendly); each
//     cluster is owned by one lane of the system's lane group and keeps its jets in that lane's VGPRs;
//   * glue - linear nodes (sums, differences, products by constants) and the state-variable
//     recursion x^[k] = rhs^[k-1] / k, which only ever need *current-order* values. These are
//     exchanged between the lanes of a group through a per-system LDS slab holding the order-k
//     coefficient of every exported u variable, and are evaluated in SIMT "rounds" driven by small
//     per-lane slot tables.
//
// Per-order flow: state-variable recursion -> clusters -> glue levels, separated by wave-level LDS
// synchronisations (a system never spans wavefronts, hence no s_barrier). The state-variable jets go to
// a small per-wave scratch (L2 / Infinity-Cache resident, laid out [order][system][variable] so that
// a wave's accesses are contiguous), are reduced for the step-size selector with wavefront shuffles,
// and are read back for the Horner / compensated update. Blocks are persistent and pull groups of
// systems from a device-side work queue, so that the scratch is proportional to the number of
// resident waves rather than to the number of systems.
//
// The arithmetic of every node is emitted by the same ssa_emitter as the unrolled mode (reference
// formulas cited in hip_emit_detail.hpp).
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <limits>
#include <map>
#include <numeric>
#include <set>

#include "hip_emit_cluster_plan.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd
{

namespace
{

using emit_detail::ssa_emitter;

using cluster_detail::cluster_plan;
using cluster_detail::glue_group;
using cluster_detail::is_var;

// u variables whose *full history* node n needs (besides, possibly, its own).
// NOTE: cu (optional) = constant u variables (constant_uvars()): a product with a constant factor is linear in the other
// one and needs no history (ssa_emitter::node evaluates it as c^[0] * x^[k]). Used by the wave-cluster planner only.
std::vector<std::uint32_t> history_operands(const dc_node &n, const std::vector<char> *cu = nullptr)
{
    std::vector<std::uint32_t> ret;
    const auto &a = n.args;
    switch (n.kind) {
        case func_kind::prod:
            if (a.size() == 2u && is_var(a[0]) && is_var(a[1])) {
                if (cu != nullptr && ((*cu)[a[0].idx] != 0 || (*cu)[a[1].idx] != 0)) {
                    break;
                }
                ret = {a[0].idx, a[1].idx};
            }
            break;
        case func_kind::div:
            if (is_var(a[1])) {
                ret = {a[1].idx};
            }
            break;
        case func_kind::sum_sq:
            for (const auto &o : a) {
                if (is_var(o)) {
                    ret.push_back(o.idx);
                }
            }
            break;
        case func_kind::pow:
            // NOTE: sqrt only needs the current order of its base, but it always needs its own history:
            // keep things simple and treat the base as a history operand in all cases.
            if (is_var(a[0])) {
                ret = {a[0].idx};
            }
            break;
        case func_kind::sin:
        case func_kind::cos:
            if (is_var(a[0])) {
                ret = {a[0].idx};
                for (const auto d : n.deps) {
                    ret.push_back(d);
                }
            }
            break;
        case func_kind::exp:
        case func_kind::log:
            if (is_var(a[0])) {
                ret = {a[0].idx};
            }
            break;
        case func_kind::tan:
        case func_kind::tanh:
        case func_kind::sinh:
        case func_kind::cosh:
        case func_kind::erf:
        case func_kind::sigmoid:
        case func_kind::asin:
        case func_kind::acos:
        case func_kind::atan:
        case func_kind::asinh:
        case func_kind::acosh:
        case func_kind::atanh:
            // Argument + hidden dependency (the functions' own histories are added by the planner).
            if (is_var(a[0])) {
                ret = {a[0].idx};
                for (const auto d : n.deps) {
                    ret.push_back(d);
                }
            }
            break;
        default:
            break;
    }
    return ret;
}

// Kinds that can be evaluated as glue (only current-order operand values).
bool is_glue_kind(const dc_node &n, const std::vector<char> *cu = nullptr)
{
    switch (n.kind) {
        case func_kind::sum:
        case func_kind::sub:
        case func_kind::num_identity:
        case func_kind::time:
            return true;
        case func_kind::prod:
            if (n.args.size() == 2u && is_var(n.args[0]) && is_var(n.args[1])) {
                return cu != nullptr && ((*cu)[n.args[0].idx] != 0 || (*cu)[n.args[1].idx] != 0);
            }
            return n.args.size() == 2u;
        case func_kind::div:
            return !is_var(n.args[1]);
        default:
            return history_operands(n, cu).empty();
    }
}

struct union_find {
    std::vector<std::uint32_t> p;
    explicit union_find(std::size_t n) : p(n)
    {
        std::iota(p.begin(), p.end(), 0u);
    }
    std::uint32_t find(std::uint32_t x)
    {
        while (p[x] != x) {
            p[x] = p[p[x]];
            x = p[x];
        }
        return x;
    }
    void unite(std::uint32_t a, std::uint32_t b)
    {
        a = find(a);
        b = find(b);
        if (a != b) {
            p[std::max(a, b)] = std::min(a, b);
        }
    }
};

} // namespace

// Build the plan; returns an empty string on success, otherwise the reason why cluster mode is not applicable.
// State variables as history operands. The planner groups a nonlinear node with the u variables whose whole history it
// needs; a *state variable* in that position (model::np1body: |r_i|^2 = sum_sq(x_i, y_i, z_i), x_i * |r_i|^-3) cannot be
// a member of a cluster. This transformation
//   - gives every state variable s read by a cluster member a glue copy g_s = sum(s),
//   - gives every state variable in history position an alias a_s = g_s - z_s (z_s: a constant-zero node of its own),
//     an ordinary cluster member with the shape of the coordinate differences of a pair cluster, and
//   - rewires the members to g_s / a_s,
// so that heliocentric and pair clusters are isomorphic and sit at the same dependency level (both read level-1 glue).
// The values are unchanged (s + nothing, s - 0). Returns false if the program needs no alias.
bool add_state_aliases(const taylor_program &p, taylor_program &out)
{
    const auto n_eq = p.n_eq;
    // Cluster members: nodes with history operands, and the u variables in history position.
    std::vector<char> member(p.n_u, 0);
    std::set<std::uint32_t> hist_sv;
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        const auto hs = history_operands(p.nodes[i]);
        if (!hs.empty()) {
            member[n_eq + i] = 1;
        }
        for (const auto h : hs) {
            if (h < n_eq) {
                hist_sv.insert(h);
            } else {
                member[h] = 1;
            }
        }
    }
    if (hist_sv.empty()) {
        return false;
    }
    // State variables read by members (in any position).
    std::set<std::uint32_t> used_sv(hist_sv.begin(), hist_sv.end());
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        if (member[n_eq + i] != 0) {
            for (const auto &o : p.nodes[i].args) {
                if (o.type == operand::kind::uvar && o.idx < n_eq) {
                    used_sv.insert(o.idx);
                }
            }
        }
    }

    out = p;
    out.nodes.clear();
    std::map<std::uint32_t, std::uint32_t> copy_of, alias_of;
    std::uint32_t next = n_eq;
    const auto uvar = [](std::uint32_t u) { return operand{operand::kind::uvar, u, 0.}; };
    for (const auto sv : used_sv) {
        dc_node g;
        g.kind = func_kind::sum;
        g.args.push_back(uvar(sv));
        out.nodes.push_back(g);
        copy_of[sv] = next++;
    }
    for (const auto sv : hist_sv) {
        dc_node z;
        z.kind = func_kind::num_identity;
        z.args.push_back(operand{operand::kind::num, 0u, 0.});
        out.nodes.push_back(z);
        dc_node a;
        a.kind = func_kind::sub;
        a.args.push_back(uvar(copy_of.at(sv)));
        a.args.push_back(uvar(next));
        out.nodes.push_back(a);
        alias_of[sv] = next + 1u;
        next += 2u;
    }
    const auto m = next - n_eq;
    const auto shift = [&](std::uint32_t u) { return u < n_eq ? u : u + m; };
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        auto nn = p.nodes[i];
        const auto hs = history_operands(p.nodes[i]);
        const bool is_member = member[n_eq + i] != 0;
        for (auto &o : nn.args) {
            if (o.type != operand::kind::uvar) {
                continue;
            }
            if (o.idx >= n_eq) {
                o.idx = shift(o.idx);
            } else if (std::find(hs.begin(), hs.end(), o.idx) != hs.end()) {
                o.idx = alias_of.at(o.idx);
            } else if (is_member) {
                o.idx = copy_of.at(o.idx);
            }
        }
        for (auto &d : nn.deps) {
            d = shift(d);
        }
        out.nodes.push_back(std::move(nn));
    }
    for (auto &d : out.sv_defs) {
        if (d.type == operand::kind::uvar) {
            d.idx = shift(d.idx);
        }
    }
    for (auto &e : out.ev_u) {
        e = shift(e);
    }
    out.n_u = p.n_u + m;
    return true;
}

namespace
{

std::string make_plan_impl(const taylor_program &p, std::uint32_t order, cluster_plan &pl,
                           const cluster_detail::plan_limits &lim);

} // namespace

// Clusters which differ by an *elided unit factor*. model::nbody() scales r^-3 by G m_j before the three products of a
// pair; when that factor is exactly 1 (unit masses in G = 1 units) the scaling node is not there and the products read the
// pow directly, while the pairs of the same system with another factor (-G m_i = -1 towards a test particle, other
// masses) keep it: the pair clusters are then neither isomorphic nor sub-shapes of each other. This pass restores the
// missing member in the INTERNAL program (the user-visible decomposition is untouched): every pow node which is read by
// products with a variable factor but by no scaling product `c * pow`, in a program where other pow nodes are read
// through such a scaling, gets `s = 1.0 * pow` right behind it, and its product readers read s (x * 1.0 is exact: no
// value changes). Returns false if there is nothing of that kind.
bool insert_unit_scalings(const taylor_program &p, taylor_program &out)
{
    const auto n_eq = p.n_eq;
    constexpr auto none = std::numeric_limits<std::uint32_t>::max();
    std::vector<char> is_pow(p.n_u, 0), scaled(p.n_u, 0), direct(p.n_u, 0);
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        const auto &n = p.nodes[u - n_eq];
        is_pow[u] = (n.kind == func_kind::pow && n.args.size() == 2u && is_var(n.args[0])) ? 1 : 0;
    }
    for (const auto &n : p.nodes) {
        if (n.kind != func_kind::prod || n.args.size() != 2u) {
            continue;
        }
        for (std::size_t a = 0; a < 2u; ++a) {
            const auto &o = n.args[a], &other = n.args[1u - a];
            if (is_var(o) && is_pow[o.idx] != 0) {
                if (other.type == operand::kind::num || other.type == operand::kind::par) {
                    scaled[o.idx] = 1;
                } else if (is_var(other)) {
                    direct[o.idx] = 1;
                }
            }
        }
    }
    bool any_scaled = false, any_todo = false;
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        any_scaled = any_scaled || scaled[u] != 0;
        any_todo = any_todo || (direct[u] != 0 && scaled[u] == 0);
    }
    if (!any_scaled || !any_todo) {
        return false;
    }
    // New index of every old u variable: one slot is inserted behind each pow node to be scaled.
    std::vector<std::uint32_t> new_idx(p.n_u, none), scal_idx(p.n_u, none);
    std::uint32_t next = n_eq;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        new_idx[i] = i;
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        new_idx[u] = next++;
        if (direct[u] != 0 && scaled[u] == 0) {
            scal_idx[u] = next++;
        }
    }
    out = p;
    out.nodes.clear();
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        auto nn = p.nodes[u - n_eq];
        const bool var_prod = nn.kind == func_kind::prod && nn.args.size() == 2u && is_var(nn.args[0]) && is_var(nn.args[1]);
        for (auto &o : nn.args) {
            if (o.type == operand::kind::uvar) {
                // (Only the products with a variable factor are rewired: other readers of the pow keep reading it.)
                o.idx = (var_prod && scal_idx[o.idx] != none) ? scal_idx[o.idx] : new_idx[o.idx];
            }
        }
        for (auto &d : nn.deps) {
            d = new_idx[d];
        }
        out.nodes.push_back(std::move(nn));
        if (scal_idx[u] != none) {
            dc_node sc;
            sc.kind = func_kind::prod;
            operand one;
            one.type = operand::kind::num;
            one.value = 1.;
            operand src;
            src.type = operand::kind::uvar;
            src.idx = new_idx[u];
            sc.args = {one, src};
            out.nodes.push_back(std::move(sc));
        }
    }
    for (auto &d : out.sv_defs) {
        if (d.type == operand::kind::uvar) {
            d.idx = new_idx[d.idx];
        }
    }
    for (auto &e : out.ev_u) {
        e = new_idx[e];
    }
    out.n_u = next;
    return true;
}

// Accelerations as flat sums of scaled pair products. model::nbody() writes the acceleration of a body as whatever its
// masses let the expression system fold: with distinct numerical masses a plain sum over the partners, each term a product
// d * (G m r^-3) or its "reaction" c * (d * (G m r^-3)) - the shape the one-lane-per-pair kernel is built for -, but with
// EQUAL masses the reactions become negations and the sums of a body turn into sum / sub / -1 * sum trees whose kind differs
// from body to body (default masses: src/model/nbody.cpp:95-130, the grouped branch), and with REPEATED masses the reactions
// are glue nodes of their own in front of the sums: two glue levels. The lanes of a glue round execute one instruction
// stream, so those systems fell to the lane-pair / first-generation kernels (1.5e8 ... 5.5e8 system-steps/s against 7.9e8).
// This pass rewrites the INTERNAL program (the user-visible decomposition is untouched): the definition of every state
// variable which is a linear tree - sum, sub, prod(number, .) - over products of two variables is flattened into
// sum_j c_j p_j; every product keeps the appearance with coefficient +1 as the direct one, the other one becomes a node
// prod(c, p) right behind it (the reaction), and the acceleration becomes ONE sum node over its terms in the order of the
// flattened tree. The result has the shape of the distinct-mass decomposition. NOTE: the additions of an acceleration are
// re-associated (one pairwise sum over all the terms instead of the tree of the expression) and nested constant factors are
// multiplied on the host: the jets of the rewritten program agree with the decomposition to rounding, not bit for bit -
// the one rewrite of this family with that property (tests: 1e3 eps on the jets of the internal program).
// Returns false if the program does not have that form or nothing would change.
bool linearise_accelerations(const taylor_program &p, taylor_program &out)
{
    const auto n_eq = p.n_eq;
    constexpr auto none = std::numeric_limits<std::uint32_t>::max();
    if (!p.ev_u.empty() || p.n_par != 0u) {
        return false;
    }
    const auto node_of = [&](std::uint32_t u) -> const dc_node & { return p.nodes[u - n_eq]; };
    const auto is_num = [](const operand &o) { return o.type == operand::kind::num; };
    // Linear glue: sum over variables, sub of two variables, prod(number, variable).
    const auto linear = [&](std::uint32_t u) {
        if (u < n_eq) {
            return false;
        }
        const auto &n = node_of(u);
        if (!n.deps.empty()) {
            return false;
        }
        if (n.kind == func_kind::sum) {
            return !n.args.empty() && std::all_of(n.args.begin(), n.args.end(), [](const operand &o) { return is_var(o); });
        }
        if (n.kind == func_kind::sub) {
            return n.args.size() == 2u && is_var(n.args[0]) && is_var(n.args[1]);
        }
        return n.kind == func_kind::prod && n.args.size() == 2u && is_num(n.args[0]) && is_var(n.args[1]);
    };
    const auto leaf_ok = [&](std::uint32_t u) {
        if (u < n_eq) {
            return false;
        }
        const auto &n = node_of(u);
        return n.kind == func_kind::prod && n.args.size() == 2u && is_var(n.args[0]) && is_var(n.args[1]) && n.deps.empty();
    };
    // Flattened definitions.
    struct term {
        double c;
        std::uint32_t leaf;
    };
    std::vector<std::vector<term>> flat(n_eq);
    std::vector<char> removed(p.n_u, 0);
    bool ok = true, any_tree = false;
    const std::function<void(std::uint32_t, double, std::vector<term> &)> walk = [&](std::uint32_t u, double c, std::vector<term> &dst) {
        if (!ok) {
            return;
        }
        if (linear(u)) {
            removed[u] = 1;
            const auto &n = node_of(u);
            if (n.kind == func_kind::sum) {
                for (const auto &o : n.args) {
                    walk(o.idx, c, dst);
                }
            } else if (n.kind == func_kind::sub) {
                walk(n.args[0].idx, c, dst);
                walk(n.args[1].idx, -c, dst);
            } else {
                walk(n.args[1].idx, c * n.args[0].value, dst);
            }
            return;
        }
        if (!leaf_ok(u)) {
            ok = false;
            return;
        }
        dst.push_back({c, u});
    };
    std::vector<std::uint32_t> acc_vars;
    for (std::uint32_t i = 0; i < n_eq && ok; ++i) {
        const auto &d = p.sv_defs[i];
        if (d.type != operand::kind::uvar) {
            return false;
        }
        if (d.idx < n_eq) {
            continue; // (x' = v)
        }
        walk(d.idx, 1., flat[i]);
        acc_vars.push_back(i);
        const auto &root = node_of(d.idx);
        // (Already one sum over leaves and reactions: nothing to do for this variable.)
        any_tree = any_tree || !(root.kind == func_kind::sum && std::all_of(root.args.begin(), root.args.end(), [&](const operand &o) {
                                     return leaf_ok(o.idx);
                                 }));
    }
    if (!ok || acc_vars.size() < 2u || !any_tree) {
        return false;
    }
    const auto n_terms = flat[acc_vars[0]].size();
    if (n_terms < 2u) {
        return false;
    }
    // Appearances of every leaf: at most two, one of them with coefficient +1 (the direct product).
    std::map<std::uint32_t, std::vector<std::pair<std::uint32_t, double>>> app;
    for (const auto i : acc_vars) {
        if (flat[i].size() != n_terms) {
            return false;
        }
        for (const auto &t : flat[i]) {
            if (!std::isfinite(t.c) || t.c == 0.) {
                return false;
            }
            app[t.leaf].emplace_back(i, t.c);
        }
    }
    std::map<std::pair<std::uint32_t, std::uint32_t>, bool> is_reaction; // (variable, leaf) -> reads the reaction node
    std::map<std::uint32_t, double> rx_coef;                            // leaf -> coefficient of its reaction node
    for (const auto &[leaf, v] : app) {
        if (v.size() > 2u || (v.size() == 2u && v[0].first == v[1].first)) {
            return false;
        }
        std::size_t direct = v.size();
        for (std::size_t a = 0; a < v.size(); ++a) {
            if (v[a].second == 1.) {
                direct = a;
                break;
            }
        }
        if (direct == v.size()) {
            return false;
        }
        for (std::size_t a = 0; a < v.size(); ++a) {
            is_reaction[{v[a].first, leaf}] = a != direct;
            if (a != direct) {
                rx_coef[leaf] = v[a].second;
            }
        }
    }
    // The removed linear nodes must have no reader outside the trees.
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (removed[u] != 0) {
            continue;
        }
        for (const auto &o : node_of(u).args) {
            if (is_var(o) && removed[o.idx] != 0) {
                return false;
            }
        }
        for (const auto dd : node_of(u).deps) {
            if (removed[dd] != 0) {
                return false;
            }
        }
    }
    // New numbering: the kept nodes in their order, a reaction node right behind its product, the sums at the end.
    std::vector<std::uint32_t> new_idx(p.n_u, none), rx_idx(p.n_u, none);
    std::uint32_t next = n_eq;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        new_idx[i] = i;
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (removed[u] != 0) {
            continue;
        }
        new_idx[u] = next++;
        if (rx_coef.count(u) != 0u) {
            rx_idx[u] = next++;
        }
    }
    out = p;
    out.nodes.clear();
    const auto uvar = [](std::uint32_t idx) {
        operand o;
        o.type = operand::kind::uvar;
        o.idx = idx;
        return o;
    };
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (removed[u] != 0) {
            continue;
        }
        auto nn = node_of(u);
        for (auto &o : nn.args) {
            if (o.type == operand::kind::uvar) {
                o.idx = new_idx[o.idx];
            }
        }
        for (auto &d : nn.deps) {
            d = new_idx[d];
        }
        out.nodes.push_back(std::move(nn));
        if (rx_idx[u] != none) {
            dc_node rx;
            rx.kind = func_kind::prod;
            operand c;
            c.type = operand::kind::num;
            c.value = rx_coef.at(u);
            rx.args = {c, uvar(new_idx[u])};
            out.nodes.push_back(std::move(rx));
        }
    }
    for (const auto i : acc_vars) {
        dc_node sm;
        sm.kind = func_kind::sum;
        for (const auto &t : flat[i]) {
            sm.args.push_back(uvar(is_reaction.at({i, t.leaf}) ? rx_idx[t.leaf] : new_idx[t.leaf]));
        }
        out.nodes.push_back(std::move(sm));
        out.sv_defs[i] = uvar(next++);
    }
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        if (p.sv_defs[i].idx < n_eq) {
            out.sv_defs[i] = p.sv_defs[i];
        }
    }
    out.n_u = next;
    return true;
}

// Distinct masses in a pair-interaction system: model::nbody() scales the power of the distance ONCE per pair
// (s = c * r^-3), multiplies the coordinate differences by it and derives the reactions from those products - scaling,
// scaled products and reactions are members of the pair's cluster. The v2 cluster phase of block mode (rolled order loop,
// rows in registers, index-pair convolutions: the kernel of model::nbody(64) with its equal masses) wants the clusters of the
// EQUAL-mass system: differences, sum of squares, power, three unit products. This pass rewrites the INTERNAL program (the
// user-visible decomposition is untouched): every product d * s reads r^-3 directly, a node c * (d * r^-3) placed right
// behind it takes its place for the sums which read it, a reaction c' * (d * s) becomes (c' c) * (d * r^-3), the scaling node
// goes. Equal to the decomposition to rounding (c (d r^-3) against d (c r^-3); the product of the two constants is formed in
// double precision). Returns false if nothing has that shape.
bool externalise_scalings(const taylor_program &p, taylor_program &out)
{
    const auto n_eq = p.n_eq;
    constexpr auto none = std::numeric_limits<std::uint32_t>::max();
    if (!p.ev_u.empty() || p.n_par != 0u) {
        return false;
    }
    const auto node_of = [&](std::uint32_t u) -> const dc_node & { return p.nodes[u - n_eq]; };
    const auto is_num = [](const operand &o) { return o.type == operand::kind::num; };
    const auto scaled = [&](const dc_node &n) {
        return n.kind == func_kind::prod && n.args.size() == 2u && is_num(n.args[0]) && is_var(n.args[1]) && n.deps.empty();
    };
    const auto product = [&](const dc_node &n) {
        return n.kind == func_kind::prod && n.args.size() == 2u && is_var(n.args[0]) && is_var(n.args[1]) && n.deps.empty();
    };
    std::vector<std::vector<std::uint32_t>> readers(p.n_u);
    std::vector<char> dep_read(p.n_u, 0);
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        for (const auto &o : node_of(u).args) {
            if (is_var(o)) {
                readers[o.idx].push_back(u);
            }
        }
        for (const auto d : node_of(u).deps) {
            dep_read[d] = 1;
        }
    }
    std::vector<char> sv_read(p.n_u, 0);
    for (const auto &d : p.sv_defs) {
        if (d.type == operand::kind::uvar) {
            sv_read[d.idx] = 1;
        }
    }
    // Scaling nodes: number * pow, read by products only.
    std::map<std::uint32_t, std::pair<double, std::uint32_t>> scal;
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        const auto &n = node_of(u);
        if (!scaled(n) || n.args[1].idx < n_eq || node_of(n.args[1].idx).kind != func_kind::pow || dep_read[u] != 0 || sv_read[u] != 0
            || readers[u].empty()) {
            continue;
        }
        if (std::all_of(readers[u].begin(), readers[u].end(), [&](std::uint32_t r) {
                const auto &rn = node_of(r);
                return product(rn) && (rn.args[0].idx == u) != (rn.args[1].idx == u);
            })) {
            scal[u] = {n.args[0].value, n.args[1].idx};
        }
    }
    if (scal.empty()) {
        return false;
    }
    // The products which read them; their readers: sums / differences, or reactions number * product.
    std::map<std::uint32_t, double> scaled_prod;
    for (const auto &[s_u, cw] : scal) {
        for (const auto pr : readers[s_u]) {
            if (dep_read[pr] != 0) {
                return false;
            }
            for (const auto r : readers[pr]) {
                const auto &rn = node_of(r);
                const bool lin = rn.deps.empty() && (rn.kind == func_kind::sum || rn.kind == func_kind::sub || scaled(rn));
                if (!lin) {
                    return false;
                }
            }
            scaled_prod[pr] = cw.first;
        }
    }
    // New numbering: the scalings go, number * product right behind every product.
    std::vector<std::uint32_t> new_idx(p.n_u, none), a_idx(p.n_u, none);
    std::uint32_t next = n_eq;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        new_idx[i] = i;
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (scal.count(u) != 0u) {
            continue;
        }
        new_idx[u] = next++;
        if (scaled_prod.count(u) != 0u) {
            a_idx[u] = next++;
        }
    }
    const auto uvar = [](std::uint32_t idx) {
        operand o;
        o.type = operand::kind::uvar;
        o.idx = idx;
        return o;
    };
    const auto num = [](double v) {
        operand o;
        o.type = operand::kind::num;
        o.value = v;
        return o;
    };
    // (What a reader of u reads now.)
    const auto reads = [&](std::uint32_t u) { return a_idx[u] != none ? a_idx[u] : new_idx[u]; };
    out = p;
    out.nodes.clear();
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (scal.count(u) != 0u) {
            continue;
        }
        auto nn = node_of(u);
        if (scaled_prod.count(u) != 0u) {
            for (auto &o : nn.args) {
                o.idx = (scal.count(o.idx) != 0u) ? new_idx[scal.at(o.idx).second] : reads(o.idx);
            }
            out.nodes.push_back(std::move(nn));
            dc_node a;
            a.kind = func_kind::prod;
            a.args = {num(scaled_prod.at(u)), uvar(new_idx[u])};
            out.nodes.push_back(std::move(a));
            continue;
        }
        if (scaled(nn) && scaled_prod.count(nn.args[1].idx) != 0u) {
            // Reaction: (c' c) * unit product.
            const auto pr = nn.args[1].idx;
            nn.args = {num(nn.args[0].value * scaled_prod.at(pr)), uvar(new_idx[pr])};
            out.nodes.push_back(std::move(nn));
            continue;
        }
        for (auto &o : nn.args) {
            if (o.type == operand::kind::uvar) {
                o.idx = reads(o.idx);
            }
        }
        for (auto &d : nn.deps) {
            d = new_idx[d];
        }
        out.nodes.push_back(std::move(nn));
    }
    for (auto &d : out.sv_defs) {
        if (d.type == operand::kind::uvar) {
            d.idx = reads(d.idx);
        }
    }
    out.n_u = next;
    return true;
}

namespace
{

bool node_of_is_pow(const taylor_program &p, std::uint32_t u)
{
    return u >= p.n_eq && p.nodes[u - p.n_eq].kind == func_kind::pow;
}

} // namespace

// Clusters which SHARE cheap linear members. The sum of squares at the head of a distance cluster reads coordinate
// differences `c - x` / `x - c` of ONE variable and a number; when several clusters have the same difference (fixed
// centres / mascons with a repeated coordinate) common-subexpression elimination gives them ONE node, and a coordinate
// which is zero turns `0 - x` into `-1 * x` (model::fixed_centres with a centre at the origin, a mascon on an axis): the
// clusters then overlap or differ in the kind of a member, and the cluster planner gives up ("clusters are not
// isomorphic": table stepper, 100x slower). This pass rewrites the INTERNAL program (the user-visible decomposition is
// untouched): every sum of squares all of whose arguments are such differences gets PRIVATE copies of them, placed right
// in front of it in argument order, negations written as `-0.0 - x` (bit for bit the same value as `-1 * x`, signed
// zeros and NaNs included: -0 - (+0) = -0, -0 - (-0) = +0); the products of the cluster (difference times the -
// possibly scaled - power of the sum of squares) read the copies; originals which nobody reads any more are dropped.
// A few redundant subtractions per step buy the cluster kernels. Returns false if nothing of that kind is found.
bool privatise_cluster_inputs(const taylor_program &p, taylor_program &out)
{
    const auto n_eq = p.n_eq;
    constexpr auto none = std::numeric_limits<std::uint32_t>::max();
    const auto node_of = [&](std::uint32_t u) -> const dc_node & { return p.nodes[u - n_eq]; };
    // Cheap linear members: sub(num, var), sub(var, num), prod(-1, var) over a STATE variable or u variable.
    const auto is_neg = [](const dc_node &n) {
        return n.kind == func_kind::prod && n.args.size() == 2u && n.args[0].type == operand::kind::num
               && n.args[0].value == -1. && is_var(n.args[1]);
    };
    const auto is_diff = [](const dc_node &n) {
        return n.kind == func_kind::sub && n.args.size() == 2u
               && ((n.args[0].type == operand::kind::num && is_var(n.args[1]))
                   || (is_var(n.args[0]) && n.args[1].type == operand::kind::num));
    };
    const auto cheap = [&](std::uint32_t u) { return u >= n_eq && (is_neg(node_of(u)) || is_diff(node_of(u))); };

    // The heads: sums of squares over cheap members only.
    std::vector<std::uint32_t> heads;
    std::vector<std::uint32_t> n_head_readers(p.n_u, 0u);
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        const auto &n = node_of(u);
        if (n.kind != func_kind::sum_sq || n.args.empty()) {
            continue;
        }
        bool ok = true;
        for (const auto &o : n.args) {
            ok = ok && is_var(o) && cheap(o.idx);
        }
        if (ok) {
            heads.push_back(u);
            for (const auto &o : n.args) {
                ++n_head_readers[o.idx];
            }
        }
    }
    if (heads.size() < 2u) {
        return false;
    }
    bool needed = false;
    bool any_diff = false;
    for (const auto h : heads) {
        for (const auto &o : node_of(h).args) {
            needed = needed || n_head_readers[o.idx] > 1u || is_neg(node_of(o.idx));
            any_diff = any_diff || is_diff(node_of(o.idx));
        }
    }
    if (!needed || !any_diff) {
        return false;
    }
    // Which head does a u variable hang off? pow(head, c) and the scalings num * pow / par * pow of it.
    std::vector<std::uint32_t> head_of(p.n_u, none);
    for (const auto h : heads) {
        head_of[h] = h;
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        const auto &n = node_of(u);
        if (n.kind == func_kind::pow && n.args.size() == 2u && is_var(n.args[0]) && head_of[n.args[0].idx] != none) {
            head_of[u] = head_of[n.args[0].idx];
        } else if (n.kind == func_kind::prod && n.args.size() == 2u && !is_var(n.args[0]) && is_var(n.args[1])
                   && node_of_is_pow(p, n.args[1].idx) && head_of[n.args[1].idx] != none) {
            head_of[u] = head_of[n.args[1].idx];
        }
    }
    // Private copies: copy_idx[(head, position)] in the new numbering; readers: the head itself and the products
    // cheap * (pow chain of the head).
    struct rewire {
        std::uint32_t node, arg, head, pos;
    };
    std::vector<rewire> rw;
    std::vector<std::uint32_t> other_readers(p.n_u, 0u);
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        const auto &n = node_of(u);
        if (head_of[u] == u) {
            continue; // (a head: handled below)
        }
        for (std::size_t a = 0; a < n.args.size(); ++a) {
            const auto &o = n.args[a];
            if (!is_var(o) || !cheap(o.idx) || n_head_readers[o.idx] == 0u) {
                continue;
            }
            bool done = false;
            if (n.kind == func_kind::prod && n.args.size() == 2u && is_var(n.args[1u - a])) {
                const auto h = head_of[n.args[1u - a].idx];
                if (h != none) {
                    const auto &ha = node_of(h).args;
                    for (std::uint32_t q = 0; q < ha.size(); ++q) {
                        if (ha[q].idx == o.idx) {
                            rw.push_back({u, static_cast<std::uint32_t>(a), h, q});
                            done = true;
                            break;
                        }
                    }
                }
            }
            if (!done) {
                ++other_readers[o.idx];
            }
        }
    }
    for (const auto &d : p.sv_defs) {
        if (d.type == operand::kind::uvar && d.idx < p.n_u) {
            ++other_readers[d.idx];
        }
    }
    for (const auto e : p.ev_u) {
        ++other_readers[e];
    }
    // New numbering: the originals which keep a reader stay where they are, the copies of a head go right in front of it.
    std::vector<std::uint32_t> new_idx(p.n_u, none);
    std::map<std::pair<std::uint32_t, std::uint32_t>, std::uint32_t> copy_idx;
    std::uint32_t next = n_eq;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        new_idx[i] = i;
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (cheap(u) && n_head_readers[u] != 0u && other_readers[u] == 0u) {
            continue; // dropped
        }
        if (head_of[u] == u) {
            for (std::uint32_t q = 0; q < node_of(u).args.size(); ++q) {
                copy_idx[{u, q}] = next++;
            }
        }
        new_idx[u] = next++;
    }
    out = p;
    out.nodes.clear();
    std::map<std::pair<std::uint32_t, std::uint32_t>, std::pair<std::uint32_t, std::uint32_t>> rw_of;
    for (const auto &r : rw) {
        rw_of[{r.node, r.arg}] = {r.head, r.pos};
    }
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        if (new_idx[u] == none) {
            continue;
        }
        auto nn = node_of(u);
        if (head_of[u] == u) {
            for (std::uint32_t q = 0; q < nn.args.size(); ++q) {
                auto c = node_of(nn.args[q].idx);
                if (is_neg(c)) {
                    // -1 * x -> -0.0 - x.
                    const auto var = c.args[1];
                    c.kind = func_kind::sub;
                    c.args[0].type = operand::kind::num;
                    c.args[0].value = -0.;
                    c.args[1] = var;
                }
                for (auto &o : c.args) {
                    if (o.type == operand::kind::uvar) {
                        o.idx = new_idx[o.idx];
                    }
                }
                out.nodes.push_back(std::move(c));
                nn.args[q].idx = copy_idx[{u, q}];
            }
        } else {
            for (std::size_t a = 0; a < nn.args.size(); ++a) {
                auto &o = nn.args[a];
                if (o.type != operand::kind::uvar) {
                    continue;
                }
                const auto it = rw_of.find({u, static_cast<std::uint32_t>(a)});
                o.idx = (it != rw_of.end()) ? copy_idx[it->second] : new_idx[o.idx];
            }
        }
        for (auto &d : nn.deps) {
            d = new_idx[d];
        }
        out.nodes.push_back(std::move(nn));
    }
    for (auto &d : out.sv_defs) {
        if (d.type == operand::kind::uvar) {
            d.idx = new_idx[d.idx];
        }
    }
    for (auto &e : out.ev_u) {
        e = new_idx[e];
    }
    out.n_u = next;
    return true;
}

// Clusters of different shapes. When every cluster is a *sub-shape* of the largest one (model::nbody with massless
// bodies: the pairs with a test particle lack the three reaction products of the massive-massive pairs), the smaller
// clusters are padded in the INTERNAL program (the user-visible decomposition is untouched, like add_state_aliases())
// with the members they lack - same kinds and operands as in the template, results nobody reads - so that all the
// clusters are isomorphic and the wave-cluster steppers apply instead of the table-driven one. The embedding of a
// cluster into the template is found by backtracking over the (small) member lists: kinds, argument structure
// (member -> member edges, consistent external inputs, structural numbers) and hidden dependencies must match.
// Returns false if the program is not of that kind.
bool pad_clusters(const taylor_program &p, std::uint32_t order, taylor_program &out)
{
    using cluster_detail::cluster_plan;
    cluster_plan pl;
    const auto why = make_plan_impl(p, order, pl, cluster_detail::plan_limits{});
    if (why.rfind("clusters are not isomorphic", 0) != 0 || pl.clusters.size() < 2u) {
        return false;
    }
    const auto n_eq = p.n_eq;
    const auto nc = pl.clusters.size();
    std::size_t tmax = 0;
    for (std::size_t c = 1; c < nc; ++c) {
        if (pl.clusters[c].size() > pl.clusters[tmax].size()) {
            tmax = c;
        }
    }
    const auto &T = pl.clusters[tmax];
    std::map<std::uint32_t, std::uint32_t> tpos;
    for (std::uint32_t q = 0; q < T.size(); ++q) {
        tpos[T[q]] = q;
    }
    // (Like the cluster signatures of make_plan(): a factor -1 is an ordinary per-cluster constant.)
    const auto structural_number = [](const dc_node &n, std::size_t a) { return n.kind == func_kind::pow && a == 1u; };
    constexpr auto none = std::numeric_limits<std::uint32_t>::max();

    // New members get provisional indices p.n_u, p.n_u + 1, ... and an anchor: the (old) u variable after which they are
    // inserted, so that the members of a padded cluster come in the order of the template (the planner compares the
    // clusters member by member, in ascending order of their indices).
    std::vector<dc_node> new_nodes;
    std::vector<std::uint32_t> anchor_of; // per new node: old u variable it follows (none: before the first node)
    std::uint32_t next_u = p.n_u;
    bool padded_any = false;
    for (std::size_t c = 0; c < nc; ++c) {
        if (c == tmax) {
            continue;
        }
        const auto &S = pl.clusters[c];
        std::map<std::uint32_t, std::uint32_t> spos;
        for (std::uint32_t q = 0; q < S.size(); ++q) {
            spos[S[q]] = q;
        }
        std::vector<std::uint32_t> f(S.size(), none);      // S position -> T position
        std::vector<char> used(T.size(), 0);
        std::map<std::uint32_t, std::uint32_t> ext_t2s, ext_s2t; // external inputs: T's u <-> S's u
        std::uint64_t budget = 200000;
        // Try to map S[i], S[i+1], ... (ascending u: the arguments of a member precede it).
        std::function<bool(std::size_t)> rec = [&](std::size_t i) -> bool {
            if (i == S.size()) {
                return true;
            }
            if (budget-- == 0u) {
                return false;
            }
            const auto &ns = p.nodes[S[i] - n_eq];
            for (std::uint32_t q = 0; q < T.size(); ++q) {
                const auto &nt = p.nodes[T[q] - n_eq];
                if (used[q] != 0 || nt.kind != ns.kind || nt.args.size() != ns.args.size() || nt.deps.size() != ns.deps.size()) {
                    continue;
                }
                bool ok = true;
                std::vector<std::pair<std::uint32_t, std::uint32_t>> added; // tentative external pairs (T u, S u)
                for (std::size_t a = 0; ok && a < ns.args.size(); ++a) {
                    const auto &so = ns.args[a], &to = nt.args[a];
                    if (so.type != to.type) {
                        ok = false;
                    } else if (is_var(so)) {
                        const auto it_t = tpos.find(to.idx);
                        const auto it_s = spos.find(so.idx);
                        if ((it_t == tpos.end()) != (it_s == spos.end())) {
                            ok = false;
                        } else if (it_t != tpos.end()) {
                            ok = f[it_s->second] == it_t->second;
                        } else {
                            const auto e1 = ext_t2s.find(to.idx);
                            const auto e2 = ext_s2t.find(so.idx);
                            if (e1 != ext_t2s.end() || e2 != ext_s2t.end()) {
                                ok = e1 != ext_t2s.end() && e2 != ext_s2t.end() && e1->second == so.idx && e2->second == to.idx;
                            } else {
                                ext_t2s[to.idx] = so.idx;
                                ext_s2t[so.idx] = to.idx;
                                added.emplace_back(to.idx, so.idx);
                            }
                        }
                    } else if (so.type == operand::kind::num) {
                        if (structural_number(nt, a) || structural_number(ns, a)) {
                            ok = so.value == to.value;
                        }
                    }
                }
                for (std::size_t d = 0; ok && d < ns.deps.size(); ++d) {
                    const auto it_t = tpos.find(nt.deps[d]);
                    const auto it_s = spos.find(ns.deps[d]);
                    ok = it_t != tpos.end() && it_s != spos.end() && f[it_s->second] == it_t->second;
                }
                if (ok) {
                    f[i] = q;
                    used[q] = 1;
                    if (rec(i + 1u)) {
                        return true;
                    }
                    used[q] = 0;
                    f[i] = none;
                }
                for (const auto &[tu, su] : added) {
                    ext_t2s.erase(tu);
                    ext_s2t.erase(su);
                }
            }
            return false;
        };
        if (!rec(0)) {
            return false;
        }
        // Members of the template the cluster lacks: appended to the program, in template order.
        std::vector<std::uint32_t> inv(T.size(), none); // T position -> u variable of this cluster
        for (std::size_t i = 0; i < S.size(); ++i) {
            inv[f[i]] = S[i];
        }
        std::uint32_t fallback_ext = none;
        for (const auto &[su, tu] : ext_s2t) {
            (void)tu;
            fallback_ext = su;
            break;
        }
        for (std::uint32_t q = 0; q < T.size(); ++q) {
            if (inv[q] != none) {
                continue;
            }
            auto nn = p.nodes[T[q] - n_eq];
            for (auto &o : nn.args) {
                if (o.type != operand::kind::uvar) {
                    continue;
                }
                if (const auto it = tpos.find(o.idx); it != tpos.end()) {
                    if (inv[it->second] == none) {
                        return false; // (template order is topological: cannot happen)
                    }
                    o.idx = inv[it->second];
                } else if (const auto e = ext_t2s.find(o.idx); e != ext_t2s.end()) {
                    o.idx = e->second;
                } else if (fallback_ext != none) {
                    o.idx = fallback_ext;
                } else {
                    return false;
                }
            }
            for (auto &d : nn.deps) {
                const auto it = tpos.find(d);
                if (it == tpos.end() || inv[it->second] == none) {
                    return false;
                }
                d = inv[it->second];
            }
            // Anchor: the member at the previous template position (or its anchor, if that one is new as well).
            std::uint32_t anc = none;
            if (q > 0u) {
                const auto prev = inv[q - 1u];
                anc = prev < p.n_u ? prev : anchor_of[prev - p.n_u];
            } else if (!S.empty() && S[0] > n_eq) {
                anc = S[0] - 1u;
            }
            new_nodes.push_back(std::move(nn));
            anchor_of.push_back(anc);
            inv[q] = next_u++;
            padded_any = true;
        }
    }
    if (!padded_any) {
        return false;
    }
    // Rebuild the program with the new members at their places; remap every u-variable index.
    std::vector<std::uint32_t> new_idx(next_u, none);
    std::vector<const dc_node *> order_nodes;
    std::vector<std::uint32_t> order_old; // provisional / old index of the nodes in the new order
    const auto place_after = [&](std::uint32_t anc) {
        for (std::size_t j = 0; j < new_nodes.size(); ++j) {
            if (anchor_of[j] == anc) {
                order_nodes.push_back(&new_nodes[j]);
                order_old.push_back(p.n_u + static_cast<std::uint32_t>(j));
            }
        }
    };
    place_after(none);
    for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
        order_nodes.push_back(&p.nodes[u - n_eq]);
        order_old.push_back(u);
        place_after(u);
    }
    // NOTE: members anchored at a state variable (a cluster whose first member directly follows the state) land first.
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        new_idx[i] = i;
    }
    for (std::size_t j = 0; j < order_old.size(); ++j) {
        new_idx[order_old[j]] = n_eq + static_cast<std::uint32_t>(j);
    }
    for (std::size_t j = 0; j < new_nodes.size(); ++j) {
        if (new_idx[p.n_u + j] == none) {
            return false; // anchored at something which is not a node (cannot happen for anchors >= n_eq)
        }
    }
    out = p;
    out.nodes.clear();
    for (const auto *np : order_nodes) {
        auto nn = *np;
        for (auto &o : nn.args) {
            if (o.type == operand::kind::uvar) {
                o.idx = new_idx[o.idx];
            }
        }
        for (auto &d : nn.deps) {
            d = new_idx[d];
        }
        out.nodes.push_back(std::move(nn));
    }
    for (auto &d : out.sv_defs) {
        if (d.type == operand::kind::uvar) {
            d.idx = new_idx[d.idx];
        }
    }
    for (auto &e : out.ev_u) {
        e = new_idx[e];
    }
    out.n_u = next_u;
    // Arguments must precede their users.
    for (std::size_t j = 0; j < out.nodes.size(); ++j) {
        for (const auto &o : out.nodes[j].args) {
            if (o.type == operand::kind::uvar && o.idx >= n_eq + j) {
                return false;
            }
        }
    }
    return true;
}

std::string cluster_detail::make_plan(const taylor_program &p, std::uint32_t order, cluster_plan &pl,
                                      const plan_limits &lim)
{
    auto why = make_plan_impl(p, order, pl, lim);
    if (lim.absorb_linear && why.rfind("clusters are not isomorphic", 0) == 0) {
        // E.g. model::np1body with unit masses: the heliocentric clusters feed scaling products of their own, the pair
        // clusters do not. Without the absorption those products are glue and the clusters are isomorphic again.
        auto lim2 = lim;
        lim2.absorb_linear = false;
        cluster_plan pl2;
        const auto why2 = make_plan_impl(p, order, pl2, lim2);
        if (why2.empty()) {
            pl = std::move(pl2);
            return why2;
        }
    }
    return why;
}

namespace
{

// Second half of make_plan_impl() for multi-class plans: the clusters are grouped into classes of identical shape and
// dependency level; constants, exported members and histories are per class; LDS slots and glue groups as usual.
std::string finish_multi_class_plan(const taylor_program &p, std::uint32_t order, cluster_plan &pl, const std::vector<char> *cu,
                                    const std::vector<std::uint32_t> &clvl, const std::vector<std::string> &sigs,
                                    const std::vector<std::vector<std::pair<std::uint32_t, std::uint32_t>>> &num_pos,
                                    const std::vector<std::vector<double>> &num_val, const std::vector<char> &exported)
{
    using cluster_detail::cluster_class;
    using cluster_detail::glue_group;
    const auto n_eq = p.n_eq, n_u = p.n_u;
    const auto nc = pl.clusters.size();
    // Classes, in the order of their first cluster.
    std::map<std::pair<std::uint32_t, std::string>, std::size_t> cidx;
    for (std::size_t c = 0; c < nc; ++c) {
        const auto key = std::make_pair(clvl[c], sigs[c]);
        auto it = cidx.find(key);
        if (it == cidx.end()) {
            it = cidx.emplace(key, pl.classes.size()).first;
            pl.classes.emplace_back();
            pl.classes.back().level = clvl[c];
        }
        pl.classes[it->second].members.push_back(static_cast<std::uint32_t>(c));
    }
    std::size_t max_members = 0, stored_total = 0;
    for (auto &cls : pl.classes) {
        const auto c0 = cls.members[0];
        const auto &t0 = pl.clusters[c0];
        max_members = std::max(max_members, cls.members.size());
        // Constants: literal if identical across the class, otherwise a per-lane table entry.
        cls.cst_val.assign(cls.members.size(), {});
        for (std::size_t j = 0; j < num_pos[c0].size(); ++j) {
            bool same = true;
            for (const auto c : cls.members) {
                const auto a = num_val[c][j], b = num_val[c0][j];
                if (!(a == b || (std::isnan(a) && std::isnan(b))) || std::signbit(a) != std::signbit(b)) {
                    same = false;
                }
            }
            if (!same) {
                cls.cst_pos.push_back(num_pos[c0][j]);
                for (std::size_t m = 0; m < cls.members.size(); ++m) {
                    cls.cst_val[m].push_back(num_val[cls.members[m]][j]);
                }
            }
        }
        for (std::uint32_t q = 0; q < t0.size(); ++q) {
            bool any = false;
            for (const auto c : cls.members) {
                any = any || exported[pl.clusters[c][q]] != 0;
            }
            if (any) {
                cls.out_pos.push_back(q);
            }
        }
        // Histories of the template.
        std::set<std::uint32_t> stored;
        for (const auto u : t0) {
            const auto &n = p.nodes[u - n_eq];
            const auto hs = history_operands(n, cu);
            for (const auto h : hs) {
                stored.insert(h);
            }
            switch (n.kind) {
                case func_kind::pow:
                case func_kind::div:
                case func_kind::sin:
                case func_kind::cos:
                case func_kind::exp:
                case func_kind::log:
                case func_kind::sigmoid:
                case func_kind::asin:
                case func_kind::acos:
                case func_kind::atan:
                case func_kind::asinh:
                case func_kind::acosh:
                case func_kind::atanh:
                    if (!hs.empty()) {
                        stored.insert(u);
                    }
                    break;
                default:
                    break;
            }
        }
        for (std::uint32_t q = 0; q < t0.size(); ++q) {
            if (stored.count(t0[q]) != 0u) {
                cls.stored_pos.push_back(q);
            }
        }
        stored_total += stored.size();
    }
    if (max_members > 64u) {
        return "more than 64 clusters in a class";
    }
    // (Every lane carries the histories of one cluster of EVERY class.)
    if (stored_total * order * 2u > 420u) {
        return "the jets of the cluster classes do not fit in the register file (" + std::to_string(stored_total) + " histories)";
    }
    pl.L = 2;
    while (pl.L < max_members && pl.L < 64u) {
        pl.L *= 2u;
    }
    pl.spw = 64u / pl.L;
    pl.max_level = *std::max_element(pl.lvl.begin(), pl.lvl.end());

    // LDS slots: state variables, exported cluster members (output-major inside a class), glue nodes.
    pl.slot_of.assign(n_u, -1);
    std::uint32_t ns = 0;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        pl.slot_of[i] = static_cast<int>(ns++);
    }
    for (const auto &cls : pl.classes) {
        for (const auto q : cls.out_pos) {
            for (const auto c : cls.members) {
                pl.slot_of[pl.clusters[c][q]] = static_cast<int>(ns++);
            }
        }
    }
    for (std::uint32_t u = n_eq; u < n_u; ++u) {
        if (pl.cluster_of[u] == -1) {
            pl.slot_of[u] = static_cast<int>(ns++);
        }
    }
    pl.n_slots = ns;

    // Glue groups: per level, per shape.
    const auto structural_number = [](const dc_node &n, std::size_t a) {
        if (n.kind == func_kind::pow && a == 1u) {
            return true;
        }
        return n.kind == func_kind::prod && a == 0u && n.args[0].type == operand::kind::num && n.args[0].value == -1.;
    };
    std::map<std::pair<std::uint32_t, std::string>, std::size_t> gidx;
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        const auto u = n_eq + i;
        if (pl.cluster_of[u] != -1) {
            continue;
        }
        const auto &n = p.nodes[i];
        std::ostringstream key;
        key << func_kind_name(n.kind);
        for (std::size_t a = 0; a < n.args.size(); ++a) {
            const auto &o = n.args[a];
            if (is_var(o)) {
                key << ((cu != nullptr && n.kind == func_kind::prod && (*cu)[o.idx] != 0) ? "k" : "v");
            } else if (o.type == operand::kind::par) {
                key << "p" << o.idx;
            } else if (structural_number(n, a)) {
                key << "n" << fp_literal(o.value);
            } else {
                key << "c";
            }
        }
        const auto k = std::make_pair(pl.lvl[u], key.str());
        auto it = gidx.find(k);
        if (it == gidx.end()) {
            it = gidx.emplace(k, pl.groups.size()).first;
            pl.groups.emplace_back();
            pl.groups.back().level = pl.lvl[u];
            pl.groups.back().key = key.str();
        }
        pl.groups[it->second].nodes.push_back(u);
    }
    std::stable_sort(pl.groups.begin(), pl.groups.end(),
                     [](const glue_group &a, const glue_group &b) { return a.level < b.level; });
    return {};
}

std::string make_plan_impl(const taylor_program &p, std::uint32_t order, cluster_plan &pl,
                           const cluster_detail::plan_limits &lim)
{
    using cluster_detail::glue_group;
    const auto n_eq = p.n_eq, n_u = p.n_u;
    for (const auto &n : p.nodes) {
        if (n.kind == func_kind::custom) {
            return "functions defined through node rules run on the straight-line and interpreted steppers";
        }
    }
    pl.n_eq = n_eq;
    pl.n_u = n_u;
    // Constant u variables: only the wave-cluster generators (ssa_emitter-based) evaluate products with a constant
    // factor linearly; the block planner keeps the general treatment.
    const auto cu_store = constant_uvars(p);
    const std::vector<char> *cu = lim.jets_in_registers ? &cu_store : nullptr;

    // 1. History edges -> clusters.
    union_find uf(n_u);
    std::vector<char> in_h(n_u, 0);
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        if (p.nodes[i].kind >= func_kind::atan2) {
            // atan2, kepE and the piecewise functions: served by the unrolled / table steppers.
            return std::string("function served by the unrolled / table steppers: ") + func_kind_name(p.nodes[i].kind);
        }
        const auto u = n_eq + i;
        const auto hs = history_operands(p.nodes[i], cu);
        for (const auto h : hs) {
            if (h < n_eq) {
                return "a state variable is a history operand of a nonlinear function";
            }
            uf.unite(u, h);
            in_h[u] = 1;
            in_h[h] = 1;
        }
    }
    pl.cluster_of.assign(n_u, -1);
    std::map<std::uint32_t, int> root_to_cluster;
    for (std::uint32_t u = n_eq; u < n_u; ++u) {
        if (in_h[u] == 0) {
            continue;
        }
        const auto r = uf.find(u);
        auto it = root_to_cluster.find(r);
        if (it == root_to_cluster.end()) {
            it = root_to_cluster.emplace(r, static_cast<int>(pl.clusters.size())).first;
            pl.clusters.emplace_back();
        }
        pl.cluster_of[u] = it->second;
    }
    if (pl.clusters.size() < 2u) {
        return "fewer than 2 clusters";
    }
    if (pl.clusters.size() > lim.max_clusters && !lim.multi_class) {
        return "more than " + std::to_string(lim.max_clusters) + " clusters";
    }

    // 2. Absorb single-source linear nodes into their source cluster.
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        const auto u = n_eq + i;
        if (!lim.absorb_linear || pl.cluster_of[u] != -1 || !is_glue_kind(p.nodes[i], cu)) {
            continue;
        }
        int src = -2;
        for (const auto &o : p.nodes[i].args) {
            if (!is_var(o)) {
                continue;
            }
            const auto c = o.idx < n_eq ? -1 : pl.cluster_of[o.idx];
            if (src == -2) {
                src = c;
            } else if (src != c) {
                src = -1;
            }
        }
        if (src >= 0) {
            pl.cluster_of[u] = src;
        }
    }
    for (std::uint32_t u = n_eq; u < n_u; ++u) {
        if (pl.cluster_of[u] >= 0) {
            pl.clusters[static_cast<std::size_t>(pl.cluster_of[u])].push_back(u);
        }
    }

    // Glue nodes must be of a kind evaluable from current-order values.
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        if (pl.cluster_of[n_eq + i] == -1 && !is_glue_kind(p.nodes[i], cu)) {
            return std::string("unsupported glue node kind: ") + func_kind_name(p.nodes[i].kind);
        }
    }

    // 3. Levels on the contracted graph. lvl of state variables = 0; a cluster runs as a unit.
    //    First pass: levels of glue nodes and provisional cluster levels, iterated to a fixed point
    //    (the graph is a DAG unless a cluster depends on itself through glue).
    const auto nc = pl.clusters.size();
    std::vector<std::uint32_t> clvl(nc, 1u);
    pl.lvl.assign(n_u, 0u);
    bool changed = true;
    std::uint32_t iters = 0;
    while (changed) {
        changed = false;
        if (++iters > n_u + 2u) {
            return "cyclic dependency between a cluster and glue nodes";
        }
        for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
            const auto u = n_eq + i;
            const auto c = pl.cluster_of[u];
            std::uint32_t need = 1;
            for (const auto &o : p.nodes[i].args) {
                if (!is_var(o)) {
                    continue;
                }
                const auto oc = o.idx < n_eq ? -1 : pl.cluster_of[o.idx];
                if (c >= 0 && oc == c) {
                    continue;
                }
                const auto ol = (oc >= 0) ? clvl[static_cast<std::size_t>(oc)] : pl.lvl[o.idx];
                need = std::max(need, ol + 1u);
            }
            if (c >= 0) {
                if (need > clvl[static_cast<std::size_t>(c)]) {
                    clvl[static_cast<std::size_t>(c)] = need;
                    changed = true;
                }
            } else if (need != pl.lvl[u]) {
                pl.lvl[u] = need;
                changed = true;
            }
        }
    }
    for (std::size_t c = 1; c < nc && !lim.multi_class; ++c) {
        if (clvl[c] != clvl[0]) {
            return "clusters at different dependency levels";
        }
    }
    pl.cluster_level = clvl[0];
    for (std::uint32_t u = n_eq; u < n_u; ++u) {
        if (pl.cluster_of[u] >= 0) {
            pl.lvl[u] = clvl[static_cast<std::size_t>(pl.cluster_of[u])];
        }
    }
    pl.max_level = *std::max_element(pl.lvl.begin(), pl.lvl.end());

    // 4. Which cluster members are read from outside their cluster (glue, other clusters, sv rules)?
    std::vector<char> exported(n_u, 0);
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        const auto c = pl.cluster_of[n_eq + i];
        for (const auto &o : p.nodes[i].args) {
            if (is_var(o) && o.idx >= n_eq && pl.cluster_of[o.idx] >= 0 && pl.cluster_of[o.idx] != c) {
                exported[o.idx] = 1;
            }
        }
    }
    for (const auto &d : p.sv_defs) {
        if (is_var(d) && d.idx >= n_eq && pl.cluster_of[d.idx] >= 0) {
            exported[d.idx] = 1;
        }
    }

    // 5. Isomorphism check + template-relative tables.
    const auto &t0 = pl.clusters[0];
    pl.ext_u.assign(nc, {});
    pl.cst_val.assign(nc, {});
    std::vector<std::string> sigs(nc);
    // Numerical operands that must be identical (they select the code shape).
    const auto structural_number = [](const dc_node &n, std::size_t a) {
        if (n.kind == func_kind::pow && a == 1u) {
            return true;
        }
        if (n.kind == func_kind::prod && a == 0u && n.args[0].type == operand::kind::num && n.args[0].value == -1.) {
            return true;
        }
        return false;
    };
    std::vector<std::vector<std::pair<std::uint32_t, std::uint32_t>>> num_pos(nc), par_pos(nc);
    std::vector<std::vector<double>> num_val(nc);
    std::vector<std::vector<std::uint32_t>> par_idx(nc);
    for (std::size_t c = 0; c < nc; ++c) {
        const auto &mem = pl.clusters[c];
        if (mem.size() != t0.size() && !lim.multi_class) {
            return "clusters are not isomorphic (different sizes)";
        }
        std::map<std::uint32_t, std::uint32_t> pos_of, ext_of;
        for (std::uint32_t q = 0; q < mem.size(); ++q) {
            pos_of[mem[q]] = q;
        }
        std::ostringstream sig;
        for (std::uint32_t q = 0; q < mem.size(); ++q) {
            const auto &n = p.nodes[mem[q] - n_eq];
            // NOTE: whether a member is read from outside is not part of the shape: a position is exported if it is in
            // any cluster (clusters padded by pad_clusters() carry members nobody reads).
            sig << func_kind_name(n.kind) << '(';
            for (std::size_t a = 0; a < n.args.size(); ++a) {
                const auto &o = n.args[a];
                if (is_var(o)) {
                    if (const auto it = pos_of.find(o.idx); it != pos_of.end()) {
                        sig << 'm' << it->second;
                    } else {
                        auto it2 = ext_of.find(o.idx);
                        if (it2 == ext_of.end()) {
                            it2 = ext_of.emplace(o.idx, static_cast<std::uint32_t>(pl.ext_u[c].size())).first;
                            pl.ext_u[c].push_back(o.idx);
                        }
                        sig << 'e' << it2->second;
                    }
                } else if (o.type == operand::kind::par) {
                    // Parameters: identical index across the clusters -> plain name, otherwise a per-lane table.
                    if (lim.generic_pars) {
                        sig << 'P';
                        par_pos[c].emplace_back(q, static_cast<std::uint32_t>(a));
                        par_idx[c].push_back(o.idx);
                    } else {
                        sig << 'p' << o.idx;
                    }
                } else if (n.kind == func_kind::pow && a == 1u) {
                    sig << 'n' << fp_literal(o.value);
                } else {
                    // NOTE: a factor -1 (a negation in the decomposition) is an ordinary constant here: -G m_i is -1 for a
                    // unit mass in G = 1 units and something else for the other bodies. Identical across the clusters ->
                    // it stays in the node (and the emitter negates); otherwise it becomes a per-lane constant and the
                    // emitter multiplies (ssa_emitter::node() checks for an override before taking the shortcut).
                    sig << 'c';
                    num_pos[c].emplace_back(q, static_cast<std::uint32_t>(a));
                    num_val[c].push_back(o.value);
                }
                sig << ',';
            }
            sig << ")[";
            for (const auto d : n.deps) {
                const auto it = pos_of.find(d);
                if (it == pos_of.end()) {
                    return "hidden dependency outside of its cluster";
                }
                sig << it->second << ',';
            }
            sig << ']';
        }
        sigs[c] = sig.str();
        if (sigs[c] != sigs[0] && !lim.multi_class) {
            return "clusters are not isomorphic";
        }
    }
    if (lim.multi_class) {
        return finish_multi_class_plan(p, order, pl, cu, clvl, sigs, num_pos, num_val, exported);
    }
    // Constants: literal if identical across clusters, otherwise a per-lane table entry.
    for (std::size_t j = 0; j < num_pos[0].size(); ++j) {
        bool same = true;
        for (std::size_t c = 1; c < nc; ++c) {
            const auto a = num_val[c][j], b = num_val[0][j];
            if (!(a == b || (std::isnan(a) && std::isnan(b))) || std::signbit(a) != std::signbit(b)) {
                same = false;
            }
        }
        if (!same) {
            pl.cst_pos.push_back(num_pos[0][j]);
            for (std::size_t c = 0; c < nc; ++c) {
                pl.cst_val[c].push_back(num_val[c][j]);
            }
        }
    }
    pl.par_idx.assign(nc, {});
    for (std::size_t j = 0; j < par_pos[0].size(); ++j) {
        bool same = true;
        for (std::size_t c = 1; c < nc; ++c) {
            same = same && par_idx[c][j] == par_idx[0][j];
        }
        if (!same) {
            pl.par_pos.push_back(par_pos[0][j]);
            for (std::size_t c = 0; c < nc; ++c) {
                pl.par_idx[c].push_back(par_idx[c][j]);
            }
        }
    }
    for (std::uint32_t q = 0; q < t0.size(); ++q) {
        bool any = false;
        for (std::size_t c = 0; c < nc; ++c) {
            any = any || exported[pl.clusters[c][q]] != 0;
        }
        if (any) {
            pl.out_pos.push_back(q);
        }
    }

    // 6. Lane-group width.
    pl.L = 2;
    while (pl.L < nc && pl.L < 64u) {
        pl.L *= 2u;
    }
    pl.spw = 64u / pl.L;

    // Register estimate: members whose history is needed keep `order` doubles alive.
    std::set<std::uint32_t> stored;
    for (const auto u : t0) {
        const auto &n = p.nodes[u - n_eq];
        for (const auto h : history_operands(n, cu)) {
            stored.insert(h);
        }
        switch (n.kind) {
            case func_kind::pow:
            case func_kind::div:
            case func_kind::sin:
            case func_kind::cos:
            case func_kind::exp:
            case func_kind::log:
            case func_kind::sigmoid:
            case func_kind::asin:
            case func_kind::acos:
            case func_kind::atan:
            case func_kind::asinh:
            case func_kind::acosh:
            case func_kind::atanh:
                if (!history_operands(n, cu).empty()) {
                    stored.insert(u);
                }
                break;
            default:
                break;
        }
    }
    if (lim.jets_in_registers && stored.size() * order * 2u > 420u) {
        return "the jets of a cluster do not fit in the register file";
    }
    for (std::uint32_t q = 0; q < t0.size(); ++q) {
        if (stored.count(t0[q]) != 0u) {
            pl.stored_pos.push_back(q);
        }
    }

    // 7. LDS slots: state variables, exported cluster members, glue nodes; then dummies.
    pl.slot_of.assign(n_u, -1);
    std::uint32_t ns = 0;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        pl.slot_of[i] = static_cast<int>(ns++);
    }
    // NOTE: output-major numbering: the lanes of a group (one cluster each) write *consecutive* slots for a
    // given output, i.e. distinct LDS banks (cluster-major numbering gave a stride of n_out doubles between
    // lanes and 2-way bank conflicts on every ds_write_b64).
    for (const auto q : pl.out_pos) {
        for (std::size_t c = 0; c < nc; ++c) {
            pl.slot_of[pl.clusters[c][q]] = static_cast<int>(ns++);
        }
    }
    for (std::uint32_t u = n_eq; u < n_u; ++u) {
        if (pl.cluster_of[u] == -1) {
            pl.slot_of[u] = static_cast<int>(ns++);
        }
    }
    pl.n_slots = ns;

    // 8. Glue groups: per level, per shape.
    std::map<std::pair<std::uint32_t, std::string>, std::size_t> gidx;
    for (std::uint32_t i = 0; i < p.nodes.size(); ++i) {
        const auto u = n_eq + i;
        if (pl.cluster_of[u] != -1) {
            continue;
        }
        const auto &n = p.nodes[i];
        std::ostringstream key;
        key << func_kind_name(n.kind);
        for (std::size_t a = 0; a < n.args.size(); ++a) {
            const auto &o = n.args[a];
            if (is_var(o)) {
                // (Constant factors of products are part of the shape: they select the linear rule.)
                key << ((cu != nullptr && n.kind == func_kind::prod && (*cu)[o.idx] != 0) ? "k" : "v");
            } else if (o.type == operand::kind::par) {
                // (Per-lane parameter tables in the pipelined generator; the first-generation one checks gpar_generic.)
                if (lim.generic_pars) {
                    key << "P";
                    pl.glue_has_par = true;
                } else {
                    key << "p" << o.idx;
                }
            } else if (structural_number(n, a)) {
                key << "n" << fp_literal(o.value);
            } else {
                key << "c";
            }
        }
        const auto k = std::make_pair(pl.lvl[u], key.str());
        auto it = gidx.find(k);
        if (it == gidx.end()) {
            it = gidx.emplace(k, pl.groups.size()).first;
            pl.groups.emplace_back();
            pl.groups.back().level = pl.lvl[u];
            pl.groups.back().key = key.str();
        }
        pl.groups[it->second].nodes.push_back(u);
    }
    std::stable_sort(pl.groups.begin(), pl.groups.end(),
                     [](const glue_group &a, const glue_group &b) { return a.level < b.level; });

    return {};
}

} // namespace

// Version 1 of the cluster kernel: separate state-variable phase, 4 LDS synchronisations per order.
// Returns a module with an empty source (and the reason in why_not) if cluster mode is not applicable.
emitted_module emit_cluster_v1(const taylor_program &p, const emit_options &opts, std::string &why_not, bool multi_class)
{
    using emit_detail::prelude;
    using emit_detail::rhofac;
    using cluster_detail::cluster_class;

    emitted_module ret;
    cluster_plan pl;
    cluster_detail::plan_limits lim;
    lim.multi_class = multi_class;
    why_not = cluster_detail::make_plan(p, opts.order, pl, lim);
    if (!why_not.empty()) {
        return ret;
    }

    const auto n_eq = p.n_eq, order = opts.order, L = pl.L, spw = pl.spw;
    const std::uint32_t bs = 256, wpb = bs / 64u;
    const auto nc = static_cast<std::uint32_t>(pl.clusters.size());
    // The classes of clusters: one section of code each (a single-class plan is the class of all the clusters).
    std::vector<cluster_class> classes = pl.classes;
    if (classes.empty()) {
        cluster_class c;
        c.members.resize(nc);
        std::iota(c.members.begin(), c.members.end(), 0u);
        c.level = pl.cluster_level;
        c.cst_pos = pl.cst_pos;
        c.cst_val = pl.cst_val;
        c.out_pos = pl.out_pos;
        c.stored_pos = pl.stored_pos;
        classes.push_back(std::move(c));
    }
    const auto sv_rounds = (n_eq + L - 1u) / L;
    const auto n_col = sv_rounds * L;

    // Dummy slots for idle lanes (they replicate real work and write to slots nobody reads).
    std::uint32_t max_round_outputs = 1u;
    for (const auto &cls : classes) {
        max_round_outputs = std::max<std::uint32_t>(max_round_outputs, static_cast<std::uint32_t>(cls.out_pos.size()));
    }
    const auto dummy_base = pl.n_slots;
    const auto n_slots_tot = pl.n_slots + max_round_outputs;
    // Pad the per-system slab to an odd number of doubles (bank spreading between the systems of a wave).
    const auto slab_stride = n_slots_tot | 1u;

    // ---- per-lane tables (unsigned short slots, double constants) ----
    std::vector<std::vector<std::uint32_t>> utbl; // each entry: L values
    std::vector<std::vector<double>> dtbl;
    const auto add_utbl = [&](std::vector<std::uint32_t> v) {
        utbl.push_back(std::move(v));
        return utbl.size() - 1u;
    };
    const auto add_dtbl = [&](std::vector<double> v) {
        dtbl.push_back(std::move(v));
        return dtbl.size() - 1u;
    };

    ssa_emitter e(p, order);
    auto &os = e.os;
    // The body is emitted into a separate stream first, because the tables it needs are only known
    // once the whole body has been generated.
    std::ostringstream body;

    // Cluster tables, per class: lane l of a system holds cluster l of the class (idle lanes replicate its first one).
    struct class_tbls {
        std::vector<std::size_t> ext_tbl, out_tbl, cst_tbl;
    };
    std::vector<class_tbls> ctb(classes.size());
    std::vector<std::pair<std::string, std::size_t>> cst_names; // (name of the per-lane constant, its double table)
    for (std::size_t ci = 0; ci < classes.size(); ++ci) {
        const auto &cls = classes[ci];
        const auto ncl = static_cast<std::uint32_t>(cls.members.size());
        const auto c0 = cls.members[0];
        const auto &t0 = pl.clusters[c0];
        const auto n_ext = static_cast<std::uint32_t>(pl.ext_u[c0].size());
        const auto cluster_at = [&](std::uint32_t l) { return cls.members[l < ncl ? l : 0u]; };
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            std::vector<std::uint32_t> v(L);
            for (std::uint32_t l = 0; l < L; ++l) {
                v[l] = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[cluster_at(l)][x]]);
            }
            ctb[ci].ext_tbl.push_back(add_utbl(std::move(v)));
        }
        for (std::uint32_t x = 0; x < cls.out_pos.size(); ++x) {
            std::vector<std::uint32_t> v(L);
            for (std::uint32_t l = 0; l < L; ++l) {
                v[l] = l < ncl ? static_cast<std::uint32_t>(pl.slot_of[pl.clusters[cls.members[l]][cls.out_pos[x]]]) : dummy_base + x;
            }
            ctb[ci].out_tbl.push_back(add_utbl(std::move(v)));
        }
        for (std::uint32_t x = 0; x < cls.cst_pos.size(); ++x) {
            std::vector<double> v(L);
            for (std::uint32_t l = 0; l < L; ++l) {
                v[l] = cls.cst_val[l < ncl ? l : 0u][x];
            }
            ctb[ci].cst_tbl.push_back(add_dtbl(std::move(v)));
            const auto [q, a] = cls.cst_pos[x];
            const auto nm = "ccst" + std::to_string(ci) + "_" + std::to_string(x);
            e.numpar_override[&p.nodes[t0[q] - n_eq].args[a]] = nm;
            cst_names.emplace_back(nm, ctb[ci].cst_tbl.back());
        }
    }

    // State-variable rounds: variable index of lane l in round r is r * L + l (clamped for padding lanes).
    struct sv_round {
        std::size_t in_tbl = 0, out_tbl = 0, cst_tbl = 0;
        bool all_var = true, any_var = false;
    };
    std::vector<sv_round> svr(sv_rounds);
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        std::vector<std::uint32_t> vin(L), vout(L);
        std::vector<double> vc(L, 0.);
        for (std::uint32_t l = 0; l < L; ++l) {
            const auto i = r * L + l;
            const auto ii = i < n_eq ? i : r * L; // padding lanes replicate the first variable of the round
            const auto &d = p.sv_defs[ii];
            if (d.type == operand::kind::uvar) {
                vin[l] = static_cast<std::uint32_t>(pl.slot_of[d.idx]);
                svr[r].any_var = true;
            } else {
                vin[l] = 0;
                svr[r].all_var = false;
                if (d.type == operand::kind::par) {
                    return ret.notes = "param", why_not = "state variable defined by a runtime parameter", ret;
                }
                vc[l] = d.value;
            }
            vout[l] = i < n_eq ? static_cast<std::uint32_t>(pl.slot_of[i]) : dummy_base;
        }
        if (svr[r].any_var && !svr[r].all_var) {
            why_not = "mixed constant / variable state-variable definitions in one round";
            return ret;
        }
        svr[r].in_tbl = add_utbl(std::move(vin));
        svr[r].out_tbl = add_utbl(std::move(vout));
        svr[r].cst_tbl = add_dtbl(std::move(vc));
    }

    // ---- emission helpers ----
    const auto sync = [&]() { os << "HY_WSYNC();\n"; };
    const auto utname = [](std::size_t t) { return "ut" + std::to_string(t); };
    const auto dtname = [](std::size_t t) { return "dt" + std::to_string(t); };

    // Glue group emission at order k: rounds over the nodes of the group.
    struct glue_round_tbls {
        std::vector<std::size_t> arg_tbl; // per arg: utbl (var) or dtbl (const) index
        std::size_t out_tbl = 0;
    };
    std::vector<std::vector<glue_round_tbls>> glue_tbls(pl.groups.size());
    for (std::size_t g = 0; g < pl.groups.size(); ++g) {
        const auto &grp = pl.groups[g];
        const auto n_nodes = static_cast<std::uint32_t>(grp.nodes.size());
        const auto rounds = (n_nodes + L - 1u) / L;
        const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
        for (std::uint32_t r = 0; r < rounds; ++r) {
            glue_round_tbls rt;
            for (std::size_t a = 0; a < n0.args.size(); ++a) {
                if (is_var(n0.args[a])) {
                    std::vector<std::uint32_t> v(L);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        const auto j = r * L + l;
                        const auto u = grp.nodes[j < n_nodes ? j : r * L];
                        v[l] = static_cast<std::uint32_t>(pl.slot_of[p.nodes[u - n_eq].args[a].idx]);
                    }
                    rt.arg_tbl.push_back(add_utbl(std::move(v)));
                } else if (n0.args[a].type == operand::kind::num) {
                    std::vector<double> v(L);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        const auto j = r * L + l;
                        const auto u = grp.nodes[j < n_nodes ? j : r * L];
                        v[l] = p.nodes[u - n_eq].args[a].value;
                    }
                    rt.arg_tbl.push_back(add_dtbl(std::move(v)));
                } else {
                    rt.arg_tbl.push_back(0);
                }
            }
            std::vector<std::uint32_t> v(L);
            for (std::uint32_t l = 0; l < L; ++l) {
                const auto j = r * L + l;
                v[l] = j < n_nodes ? static_cast<std::uint32_t>(pl.slot_of[grp.nodes[j]]) : dummy_base;
            }
            rt.out_tbl = add_utbl(std::move(v));
            glue_tbls[g].push_back(std::move(rt));
        }
    }

    // Emit one glue group at order k by temporarily aliasing the first node of each round: the node
    // rule is emitted once per round with operands read from the slab through the lane's tables.
    // Constant operands (constant_uvars()): the value a round has read at order 0 is the one the linear rule
    // c^[0] * x^[k] of ssa_emitter::node() uses at every order (same position in all the nodes of a group: the shape key
    // of a group does not distinguish them, so const-ness is checked over the whole group).
    std::map<std::tuple<std::size_t, std::size_t, std::size_t>, std::string> glue_c0;
    const auto emit_glue_group = [&](std::size_t g, std::uint32_t k) {
        const auto &grp = pl.groups[g];
        const auto rep = grp.nodes[0];
        const auto &n0 = p.nodes[rep - n_eq];
        for (std::size_t r = 0; r < glue_tbls[g].size(); ++r) {
            const auto &rt = glue_tbls[g][r];
            std::map<const operand *, std::string> saved = e.numpar_override;
            std::vector<std::pair<std::uint32_t, std::string>> saved_vals;
            std::vector<std::pair<std::uint32_t, std::string>> saved_vals0;
            for (std::size_t a = 0; a < n0.args.size(); ++a) {
                const auto &o = n0.args[a];
                if (is_var(o)) {
                    const auto nm = e.def("slab[" + utname(rt.arg_tbl[a]) + "]");
                    saved_vals.emplace_back(o.idx, e.val(o.idx, k));
                    e.val(o.idx, k) = nm;
                    if (e.cu[o.idx] != 0) {
                        if (k == 0u) {
                            glue_c0[{g, r, a}] = nm;
                        } else {
                            saved_vals0.emplace_back(o.idx, e.val(o.idx, 0));
                            e.val(o.idx, 0) = glue_c0.at({g, r, a});
                        }
                    }
                } else if (o.type == operand::kind::num) {
                    e.numpar_override[&o] = dtname(rt.arg_tbl[a]);
                }
            }
            // NOTE: structural numbers (-1 of a negation) keep their literal value; constants are
            // read from the per-lane table only at the orders where the rule uses them.
            if (n0.kind == func_kind::prod && n0.args[0].type == operand::kind::num && n0.args[0].value == -1.) {
                e.numpar_override.erase(&n0.args[0]);
            }
            e.node(rep - n_eq, k);
            os << "slab[" << utname(rt.out_tbl) << "] = " << e.val(rep, k) << ";\n";
            // Restore.
            for (auto it = saved_vals.rbegin(); it != saved_vals.rend(); ++it) {
                e.val(it->first, k) = it->second;
            }
            for (auto it = saved_vals0.rbegin(); it != saved_vals0.rend(); ++it) {
                e.val(it->first, 0) = it->second;
            }
            e.numpar_override = std::move(saved);
        }
    };

    // Code of class ci at order k (the template cluster of the class; the lanes differ by their slot tables / constants).
    const auto emit_cluster = [&](std::size_t ci, std::uint32_t k) {
        const auto &cls = classes[ci];
        const auto c0 = cls.members[0];
        const auto &t0 = pl.clusters[c0];
        for (std::size_t x = 0; x < ctb[ci].ext_tbl.size(); ++x) {
            e.val(pl.ext_u[c0][x], k) = e.def("slab[" + utname(ctb[ci].ext_tbl[x]) + "]");
        }
        for (const auto u : t0) {
            e.node(u - n_eq, k);
        }
        for (std::size_t x = 0; x < cls.out_pos.size(); ++x) {
            os << "slab[" << utname(ctb[ci].out_tbl[x]) << "] = " << e.val(t0[cls.out_pos[x]], k) << ";\n";
        }
    };

    // All the levels of order k (k < order).
    const auto emit_levels = [&](std::uint32_t k) {
        for (std::uint32_t lev = 1; lev <= pl.max_level; ++lev) {
            for (std::size_t ci = 0; ci < classes.size(); ++ci) {
                if (classes[ci].level == lev) {
                    emit_cluster(ci, k);
                }
            }
            for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                if (pl.groups[g].level == lev) {
                    emit_glue_group(g, k);
                }
            }
            sync();
        }
    };

    const auto jet_off = [&](std::uint32_t k, std::uint32_t r) {
        // jetw[(k * spw + q) * n_col + r * L + l]
        return "jetl[" + std::to_string(static_cast<std::uint64_t>(k) * spw * n_col + static_cast<std::uint64_t>(r) * L)
               + "]";
    };

    // ===================== kernel body =====================
    // Order 0: publish the state.
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        os << "slab[" << utname(svr[r].out_tbl) << "] = xs" << r << ";\n";
        os << jet_off(0, r) << " = xs" << r << ";\n";
    }
    {
        std::string m = "fabs(xs0)";
        os << "double m0 = fabs(xs0);\n";
        for (std::uint32_t r = 1; r < sv_rounds; ++r) {
            os << "if (svalid" << r << ") m0 = hy_max(m0, fabs(xs" << r << "));\n";
        }
        (void)m;
    }
    sync();
    emit_levels(0);
    os << "double mo = 0.0, mom1 = 0.0;\n";
    for (std::uint32_t k = 1; k <= order; ++k) {
        // State-variable recursion: read rhs^[k-1], sync, write x^[k].
        std::vector<std::string> rhs(sv_rounds);
        for (std::uint32_t r = 0; r < sv_rounds; ++r) {
            if (svr[r].any_var) {
                rhs[r] = e.def("slab[" + utname(svr[r].in_tbl) + "]");
            }
        }
        sync();
        for (std::uint32_t r = 0; r < sv_rounds; ++r) {
            std::string xv;
            if (svr[r].any_var) {
                xv = e.def(rhs[r] + " / " + fp_literal(static_cast<double>(k)));
            } else {
                xv = (k == 1u) ? dtname(svr[r].cst_tbl) : std::string("0.0");
            }
            os << "slab[" << utname(svr[r].out_tbl) << "] = " << xv << ";\n";
            os << jet_off(k, r) << " = " << xv << ";\n";
            if (k == order - 1u || k == order) {
                const char *acc = (k == order) ? "mo" : "mom1";
                if (r == 0u) {
                    os << acc << " = fabs(" << xv << ");\n";
                } else {
                    os << "if (svalid" << r << ") " << acc << " = hy_max(" << acc << ", fabs(" << xv << "));\n";
                }
            }
        }
        sync();
        if (k < order) {
            emit_levels(k);
        }
    }
    body << os.str();
    os.str("");
    os.clear();

    // ===================== module text =====================
    std::ostringstream src;
    src << prelude;
    emit_detail::emit_dout(src, p, opts);

    src << emit_detail::wsync_macro;
    src << "__constant__ unsigned short hy_utbl[" << std::max<std::size_t>(utbl.size(), 1u) * L << "] = {";
    for (const auto &v : utbl) {
        for (const auto x : v) {
            src << x << ",";
        }
    }
    src << "};\n";
    src << "__constant__ double hy_dtbl[" << std::max<std::size_t>(dtbl.size(), 1u) * L << "] = {";
    for (const auto &v : dtbl) {
        for (const auto x : v) {
            src << fp_literal(x) << ",";
        }
    }
    src << "};\n";

    src << "extern \"C\" __global__ void __launch_bounds__(" << bs << ") hy_taylor(const hy_kargs a)\n{\n";
    src << "__shared__ double lds_slab[" << static_cast<std::uint64_t>(wpb) * spw * slab_stride << "];\n";
    src << "const unsigned lane = threadIdx.x & 63u;\nconst unsigned wib = threadIdx.x >> 6;\n";
    src << "const unsigned l = lane % " << L << "u;\nconst unsigned q = lane / " << L << "u;\n";
    src << "const u64 N = a.N;\n";
    src << "double *const slab = lds_slab + (wib * " << spw << "u + q) * " << slab_stride << "u;\n";
    src << "const u64 gwave = (u64)blockIdx.x * " << wpb << "u + wib;\n";
    // The jets of the state variables of the systems of a wavefront ([order][system][variable]): in LDS when they fit next to
    // the slab (round 6: one workgroup per CU runs anyway - one wavefront per SIMD -, and through global scratch they cost
    // the chain of 16 pendula 44 GB of HBM traffic per launch, 1.2 TB/s), otherwise in a per-wavefront global scratch.
    const std::uint64_t jet_doubles_per_wave = static_cast<std::uint64_t>(order + 1u) * spw * n_col;
    const bool jets_in_lds
        = (static_cast<std::uint64_t>(wpb) * spw * slab_stride + static_cast<std::uint64_t>(wpb) * jet_doubles_per_wave) * 8u + 2048u
          <= 160u * 1024u;
    if (jets_in_lds) {
        src << "__shared__ double lds_jets[" << static_cast<std::uint64_t>(wpb) * jet_doubles_per_wave << "];\n";
        src << "double *const jetw = lds_jets + wib * " << jet_doubles_per_wave << "u;\n";
    } else {
        src << "double *const jetw = a.scratch + gwave * " << jet_doubles_per_wave << "ull;\n";
    }
    src << "double *const jetl = jetw + q * " << n_col << "u + l;\n";
    // Per-lane table entries (loop invariant).
    for (std::size_t t = 0; t < utbl.size(); ++t) {
        src << "const unsigned ut" << t << " = hy_utbl[" << t * L << "u + l];\n";
    }
    for (std::size_t t = 0; t < dtbl.size(); ++t) {
        src << "const double dt" << t << " = hy_dtbl[" << t * L << "u + l];\n";
    }
    for (const auto &[nm, t] : cst_names) {
        src << "const double " << nm << " = dt" << t << ";\n";
    }
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        src << "const bool svalid" << r << " = (" << r * L << "u + l) < " << n_eq << "u;\n";
        src << "const unsigned svi" << r << " = svalid" << r << " ? (" << r * L << "u + l) : " << r * L << "u;\n";
    }
    src << R"HIP(
for (;;) {
// Pull the next group of systems from the device-side work queue.
u64 base = 0;
if (lane == 0u) base = atomicAdd((u64 *)(a.counters + 2), (u64)SPW);
base = __shfl(base, 0, 64);
if (base >= N) break;
// NOTE: lanes beyond the end of the ensemble replicate the last system (no side effects).
const bool live = (base + q) < N;
const u64 s = live ? (base + q) : (N - 1u);
double t_hi = a.time_hi[s], t_lo = a.time_lo[s];
)HIP";
    for (std::uint32_t i = 0; i < p.n_par; ++i) {
        src << "const double par_" << i << " = a.pars[(u64)" << i << "u * N + s];\n";
    }
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        src << "double xs" << r << " = a.state[(u64)svi" << r << " * N + s];\n";
    }
    src << R"HIP(
hy_df tfin, rem;
tfin.hi = 0.0; tfin.lo = 0.0; rem.hi = 0.0; rem.lo = 0.0;
bool t_dir = true;
double mdt = __builtin_inf();
double step_lim = 0.0;
if (a.mode == 1) {
    tfin.hi = (a.tfin_hi != nullptr) ? a.tfin_hi[s] : a.tfin_s_hi;
    tfin.lo = (a.tfin_hi != nullptr) ? a.tfin_lo[s] : a.tfin_s_lo;
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    rem = hy_df_sub(tfin, tcur);
    t_dir = (rem.hi > 0.0) || (rem.hi == 0.0 && rem.lo >= 0.0);
    if (a.lim != nullptr) mdt = a.lim[s];
} else {
    step_lim = a.lim[s];
}
u64 n_steps = 0, iter = 0;
double min_h = __builtin_inf(), max_h = 0.0, last_h = 0.0;
i64 outcome = HY_OC_SUCCESS;
for (;;) {
double lim;
if (a.mode == 1) {
    hy_df m; m.lo = 0.0;
    // NOTE: selects, not an if/else on the (per-lane) direction: see the note on HY_LIBM1.
    m.hi = t_dir ? mdt : -mdt;
    const bool lt_fwd = hy_df_lt(rem, m), lt_bwd = hy_df_lt(m, rem);
    const bool rem_first = (t_dir & lt_fwd) | (!t_dir & lt_bwd);
    lim = rem_first ? rem.hi : m.hi;
} else {
    lim = step_lim;
}
)HIP";
    src << body.str();

    // Step size: reduce the partial maxima over the lanes of the group with wavefront shuffles.
    for (std::uint32_t m = 1; m < L; m *= 2u) {
        src << "m0 = hy_max(m0, __shfl_xor(m0, " << m << ", 64));\n";
        src << "mo = hy_max(mo, __shfl_xor(mo, " << m << ", 64));\n";
        src << "mom1 = hy_max(mom1, __shfl_xor(mom1, " << m << ", 64));\n";
    }
    src << "const double num_rho = (m0 <= 1.0) ? 1.0 : m0;\n";
    src << "const double rho_o = hy_root(num_rho / mo, " << fp_literal(1. / static_cast<double>(order)) << ");\n";
    src << "const double rho_om1 = hy_root(num_rho / mom1, " << fp_literal(1. / static_cast<double>(order - 1u))
        << ");\n";
    src << "const double rho_m = hy_min(rho_o, rho_om1);\n";
    src << "double h = rho_m * " << fp_literal(rhofac(order)) << ";\n";
    src << "h = hy_min(h, fabs(lim));\nh = (lim < 0.0) ? -h : h;\n";

    // State update from the scratch jets.
    src << "asm volatile(\"\" ::: \"memory\");\n";
    const auto kstride = static_cast<std::uint64_t>(spw) * n_col;
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        src << "{\nconst double *c = jetl + " << static_cast<std::uint64_t>(r) * L << "u;\n";
        if (opts.high_accuracy) {
            src << "double res = c[0], comp = 0.0, cur_h = h;\n#pragma unroll\n";
            src << "for (unsigned k = 1; k <= " << order << "u; ++k) {\n";
            src << "const double tmp = c[(u64)k * " << kstride << "u] * cur_h;\nconst double y = tmp - comp;\n";
            src << "const double t = res + y;\ncomp = (t - res) - y;\nres = t;\ncur_h = cur_h * h;\n}\n";
        } else {
            src << "double res = c[(u64)" << order << "u * " << kstride << "u];\n#pragma unroll\n";
            src << "for (unsigned k = 1; k <= " << order << "u; ++k) {\n";
            src << "res = c[(u64)(" << order << "u - k) * " << kstride << "u] + res * h;\n}\n";
        }
        src << "xs" << r << " = res;\n}\n";
    }
    src << R"HIP(
{
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    hy_df hh; hh.hi = h; hh.lo = 0.0;
    const hy_df nt = hy_df_add(tcur, hh);
    t_hi = nt.hi; t_lo = nt.lo;
}
last_h = h;
int nfi = !(hy_finite(t_hi) && hy_finite(t_lo)) ? 1 : 0;
)HIP";
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        src << "if (svalid" << r << " && !hy_finite(xs" << r << ")) nfi = 1;\n";
    }
    for (std::uint32_t m = 1; m < L; m *= 2u) {
        src << "nfi |= __shfl_xor(nfi, " << m << ", 64);\n";
    }
    src << R"HIP(
// The coefficients of the step just taken, in the reference's layout, if requested.
if (a.tc != nullptr && live) {
)HIP";
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        src << "if (svalid" << r << ") {\nconst double *c = jetl + " << static_cast<std::uint64_t>(r) * L
            << "u;\nfor (unsigned k = 0; k <= " << order << "u; ++k) a.tc[((u64)svi" << r << " * "
            << (order + 1u) << "u + k) * N + s] = c[(u64)k * " << kstride << "u];\n}\n";
    }
    src << R"HIP(
}
HY_STEP_TAIL(nfi != 0, l == 0u && live)
}
)HIP";
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        src << "if (svalid" << r << " && live) a.state[(u64)svi" << r << " * N + s] = xs" << r << ";\n";
    }
    src << R"HIP(
if (l == 0u && live) {
    if (a.mode != 2) {
        a.time_hi[s] = t_hi;
        a.time_lo[s] = t_lo;
    } else {
        const_cast<double *>(a.lim)[s] = last_h;
    }
    a.last_h[s] = last_h;
    a.outcome[s] = outcome;
    if (a.mode == 1) {
        a.min_h[s] = min_h;
        a.max_h[s] = max_h;
        a.n_steps[s] = n_steps;
    }
}
}
}
)HIP";

    auto text = src.str();
    // SPW macro used by the work-queue code.
    const std::string spw_def = "#define SPW " + std::to_string(spw) + "u\n";
    ret.source = spw_def + text;
    ret.kernel_name = "hy_taylor";
    ret.dout_name = "hy_dout";
    ret.block_size = bs;
    ret.lanes_per_system = L;
    ret.lds_bytes = 0;
    ret.mode = emit_mode::cluster;
    ret.n_statements = e.n_stmt;
    ret.scratch_per_wave = jets_in_lds ? 0u : jet_doubles_per_wave;
    ret.persistent = true;
    ret.tc_optional = true;
    if (classes.size() == 1u) {
        ret.notes = "cluster mode: " + std::to_string(nc) + " clusters of " + std::to_string(pl.clusters[0].size()) + " nodes";
    } else {
        ret.notes = "cluster mode (" + std::to_string(classes.size()) + " classes of clusters:";
        for (const auto &cls : classes) {
            ret.notes += " " + std::to_string(cls.members.size()) + " x " + std::to_string(pl.clusters[cls.members[0]].size())
                         + " nodes at level " + std::to_string(cls.level) + ",";
        }
        ret.notes.back() = ')';
        ret.notes += ": " + std::to_string(nc) + " clusters";
    }
    ret.notes += ", L=" + std::to_string(L) + ", " + std::to_string(pl.n_slots) + " LDS slots, "
                 + std::to_string(pl.groups.size()) + " glue groups, " + std::to_string(utbl.size()) + " slot tables, jets in "
                 + (jets_in_lds ? "LDS" : "global scratch");
    (void)n_slots_tot;
    return ret;
}

//V2_PLACEHOLDER

emitted_module emit_cluster_v2(const taylor_program &p, const emit_options &opts, std::string &why_not);

// Returns a module with an empty source (and the reason in why_not) if cluster mode is not applicable.
emitted_module emit_cluster_or_empty(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    if (!opts.dev.cluster_v1 && opts.cluster_kernel != 1) {
        std::string why2;
        auto m = emit_cluster_v2(p, opts, why2);
        if (!m.source.empty()) {
            return m;
        }
        auto m1 = emit_cluster_v1(p, opts, why_not, false);
        if (!m1.source.empty()) {
            m1.notes += " (pipelined cluster kernel not applicable: " + why2 + ")";
        }
        return m1;
    }
    return emit_cluster_v1(p, opts, why_not, false);
}

// Clusters of several shapes / at several dependency levels: one section of code per class (see cluster_class).
emitted_module emit_cluster_multi_or_empty(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    return emit_cluster_v1(p, opts, why_not, true);
}

} // namespace heyoka_amd
