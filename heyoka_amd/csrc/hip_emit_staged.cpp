// "Staged" variant of the table code generation mode: the tape of normalised derivatives of ONE system lives in LDS for
// the whole step, the lanes of a workgroup are spread over the NODES of the decomposition, and the code is specialised
// per group of nodes.
//
// The reference's compact mode (taylor_compute_jet_compact_mode(), src/taylor_02.cpp:1194-1260) partitions the
// decomposition into segments (taylor_segment_dc(), :105-207: the u variables of a segment do not depend on each other),
// groups the nodes of a segment which call the same function and runs every group as a loop over index tables
// (:983-1189); its tape is a memory buffer tape[k * n_u + u] of SIMD vectors. Here:
//
//   * segment            -> dependency level with respect to the SAME-ORDER operands;
//   * (segment, function) -> group: level + elementary function + kinds of the arguments (u variable / number /
//                           parameter) + the numbers which select a formula (exponent of pow, factor -1 of a product);
//   * loop over the group -> the lanes of the workgroup, round by round: lane l of round r evaluates node r * LANES + l
//                           of the group; the tape rows of its operands and its constants sit in REGISTERS of the lane
//                           (loaded once per kernel from tables in the module), so the code of a group is the rule of
//                           its function with LDS offsets in place of the reference's index tables;
//   * tape               -> hy_lds_tape[u * HY_P + k] in LDS, u-major, odd row length (the lanes of a group read rows of
//                           different u variables at the same order: different banks); orders are a ROLLED loop like in
//                           the reference (code size and compile time independent of the order);
//   * SIMD batch         -> one system per workgroup of 64 / 128 / 256 lanes (as many workgroups per CU as tapes fit in
//                           its 160 KB of LDS), persistent workgroups pulling systems from a device-side queue.
//
// The arithmetic of every rule is the one of the interpreted stepper (hip_emit_table.cpp, hy_diff_*): running sums in
// increasing j inside the convolutions (src/math/prod.cpp:686-698), pairwise sums over the arguments of sum / sum_sq
// (src/math/sum.cpp:355), true divisions - built without FMA contraction it reproduces the compact-mode oracle bit for
// bit. Functions without a specialised rule here (tan ... atanh, atan2, kepE, the piecewise functions, node rules, rare
// argument shapes) go through the interpreter's hy_node_value() with the node index in a register.
//
// What the tape in LDS buys: a convolution term costs two LDS reads instead of two HBM / L2 round trips (the one-lane-
// per-system variant is bound by the latency of its HBM tape: every coefficient is re-read O(order) times).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "hip_emit_detail.hpp"

namespace heyoka_amd
{

namespace table_detail
{
void emit_tables_and_rules(std::ostream &src, const taylor_program &p, const emit_options &opts, const std::string &pre_defs,
                           const char *stride);
}

namespace
{

// LDS of a CU (gfx950) minus what the kernel needs besides the tape (reduction scratch, queue slot).
constexpr std::uint64_t lds_per_cu = 160u * 1024u;
constexpr std::uint64_t lds_reserve = 256u;

bool is_uvar(const operand &o)
{
    return o.type == operand::kind::uvar;
}

// Code shape of a group: a specialised rule, or the interpreter.
enum class spec { sum, sub, prod, div, sum_sq, pow, sin, cos, exp, log, time, number, generic };

struct group {
    std::uint32_t level = 0;
    spec sp = spec::generic;
    std::string key;
    std::vector<std::uint32_t> nodes; // node indices (u variable n_eq + i)
};

// Per-lane values of the generated kernel: a named register per (table row); row = one value per lane of the workgroup.
struct lane_tables {
    std::uint32_t lanes = 64;
    std::vector<std::vector<std::uint32_t>> urows;
    std::vector<std::vector<double>> drows;
    // Parameter values: per lane the index of a parameter, loaded from a.pars when a system is picked up.
    std::vector<std::vector<std::uint32_t>> prows;

    std::string add_u(std::vector<std::uint32_t> v)
    {
        urows.push_back(std::move(v));
        return "ru" + std::to_string(urows.size() - 1u);
    }
    std::string add_d(std::vector<double> v)
    {
        drows.push_back(std::move(v));
        return "rd" + std::to_string(drows.size() - 1u);
    }
    std::string add_p(std::vector<std::uint32_t> v)
    {
        prows.push_back(std::move(v));
        return "rp" + std::to_string(prows.size() - 1u);
    }
};

// pairwise_sum() of the reference over a list of expressions (adjacent pairs, a term without partner moves up).
std::string pairwise(std::vector<std::string> t)
{
    while (t.size() > 1u) {
        std::vector<std::string> n;
        for (std::size_t i = 0; i + 1u < t.size(); i += 2u) {
            n.push_back("(" + t[i] + " + " + t[i + 1u] + ")");
        }
        if (t.size() % 2u == 1u) {
            n.push_back(t.back());
        }
        t = std::move(n);
    }
    return t[0];
}

} // namespace

emitted_module emit_staged(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    emitted_module ret;
    const auto n_eq = p.n_eq, n_u = p.n_u, order = opts.order;
    // Row length: order + 1 coefficients, made odd.
    const std::uint32_t P = (order + 1u) | 1u;
    // Strict order of the additions (kw::compact_mode / kw::sum_order = running: the reference's compact-mode arithmetic,
    // bit for bit without FMA contraction): one lane per convolution. Otherwise the terms of a convolution are dealt to
    // 2 or 4 adjacent lanes (term j to lane j mod s) whose partial sums are added pairwise through DPP: the same terms in
    // another order of addition.
    const bool strict = opts.sum_order == 2;

    // ---- levels and groups ----
    const auto n_nodes = static_cast<std::uint32_t>(p.nodes.size());
    std::vector<std::uint32_t> level(n_nodes, 0u);
    std::uint32_t n_levels = 0;
    for (std::uint32_t i = 0; i < n_nodes; ++i) {
        std::uint32_t l = 0;
        for (const auto &o : p.nodes[i].args) {
            if (is_uvar(o) && o.idx >= n_eq) {
                l = std::max(l, level[o.idx - n_eq] + 1u);
            }
        }
        level[i] = l;
        n_levels = std::max(n_levels, l + 1u);
    }
    const auto arg_code = [](const operand &o) { return is_uvar(o) ? 'v' : (o.type == operand::kind::par ? 'p' : 'n'); };
    const auto classify = [&](const dc_node &n, std::string &key) {
        std::ostringstream k;
        k << func_kind_name(n.kind) << ':';
        for (const auto &o : n.args) {
            k << arg_code(o);
        }
        const auto all_v = [&]() {
            return std::all_of(n.args.begin(), n.args.end(), [](const operand &o) { return is_uvar(o); });
        };
        spec s = spec::generic;
        switch (n.kind) {
            case func_kind::sum:
                s = n.args.size() <= 8u && !n.args.empty() ? spec::sum : spec::generic;
                break;
            case func_kind::sub:
                s = n.args.size() == 2u ? spec::sub : spec::generic;
                break;
            case func_kind::prod:
                if (n.args.size() == 2u) {
                    s = spec::prod;
                    // (A leading number -1 makes the product a negation: src/math/prod.cpp:366-368.)
                    if (n.args[0].type == operand::kind::num && n.args[0].value == -1.) {
                        k << ":neg";
                    }
                }
                break;
            case func_kind::div:
                s = n.args.size() == 2u ? spec::div : spec::generic;
                break;
            case func_kind::sum_sq:
                s = (all_v() && n.args.size() <= 8u && !n.args.empty()) ? spec::sum_sq : spec::generic;
                break;
            case func_kind::pow:
                if (n.args.size() == 2u && is_uvar(n.args[0]) && n.args[1].type == operand::kind::num) {
                    s = spec::pow;
                    k << ':' << fp_literal(n.args[1].value);
                }
                break;
            case func_kind::sin:
                s = (all_v() && n.deps.size() == 1u) ? spec::sin : spec::generic;
                break;
            case func_kind::cos:
                s = (all_v() && n.deps.size() == 1u) ? spec::cos : spec::generic;
                break;
            case func_kind::exp:
                s = all_v() ? spec::exp : spec::generic;
                break;
            case func_kind::log:
                s = all_v() ? spec::log : spec::generic;
                break;
            case func_kind::time:
                s = spec::time;
                break;
            case func_kind::num_identity:
                s = (n.args.size() == 1u && !is_uvar(n.args[0])) ? spec::number : spec::generic;
                break;
            default:
                break;
        }
        if (s == spec::generic) {
            // One group per level for everything the interpreter serves (the rule is selected per lane at run time).
            key = "generic";
        } else {
            key = k.str();
        }
        return s;
    };
    // Which u variables need their whole HISTORY on the tape (read at orders below the current one: operands of
    // convolutions, functions with a recurrence on themselves, everything the interpreter touches, the state variables -
    // Taylor coefficients and update - and the event equations), and which are only ever read at the current order: those
    // get ONE cell ("slab") which every order overwrites. For the outer Solar System 126 of 234 u variables need a row:
    // 22 KB instead of 39 KB per system, 7 systems per CU instead of 4.
    std::vector<spec> spec_of(n_nodes);
    std::vector<std::string> key_of(n_nodes);
    for (std::uint32_t i = 0; i < n_nodes; ++i) {
        spec_of[i] = classify(p.nodes[i], key_of[i]);
    }
    // Virtual differences. A difference d = a - b of two u variables which is read ONLY by convolutions (the coordinate
    // differences of an N-body system: operands of a sum of squares and of the products d r^-3) needs no row of its own:
    // its consumers subtract on the fly, d^[j] = a^[j] - b^[j] - the same operation on the same operands, the same value bit
    // for bit. 45 of the 126 rows of the outer Solar System go (10 systems per CU instead of 7: the throughput of this
    // stepper is systems in flight over the latency of a step) and the level of the differences with them.
    // (HEYOKA_AMD_TABLE_LDS=6 switches it off - A/B.)
    std::vector<char> virt(n_nodes, 0);
    if (opts.dev.table_lds != 6 && opts.dev.table_lds != 3) {
        std::vector<char> ok_virt(n_nodes, 0);
        for (std::uint32_t i = 0; i < n_nodes; ++i) {
            const auto &n = p.nodes[i];
            ok_virt[i] = spec_of[i] == spec::sub && n.args.size() == 2u && is_uvar(n.args[0]) && is_uvar(n.args[1]);
        }
        for (const auto u : p.ev_u) {
            if (u >= n_eq) {
                ok_virt[u - n_eq] = 0;
            }
        }
        for (const auto &d : p.sv_defs) {
            if (is_uvar(d) && d.idx >= n_eq) {
                ok_virt[d.idx - n_eq] = 0;
            }
        }
        std::vector<char> used(n_nodes, 0);
        for (std::uint32_t i = 0; i < n_nodes; ++i) {
            const auto &n = p.nodes[i];
            const bool conv_reader = (spec_of[i] == spec::sum_sq)
                                     || (spec_of[i] == spec::prod && n.args.size() == 2u && is_uvar(n.args[0]) && is_uvar(n.args[1]));
            for (const auto &o : n.args) {
                if (is_uvar(o) && o.idx >= n_eq) {
                    used[o.idx - n_eq] = 1;
                    if (!conv_reader) {
                        ok_virt[o.idx - n_eq] = 0;
                    }
                }
            }
            for (const auto d : n.deps) {
                if (d >= n_eq) {
                    ok_virt[d - n_eq] = 0;
                }
            }
            // (A product of two virtual differences, or a sum of squares with a mix of virtual and stored arguments, keeps
            // the code simple by storing: decided below, once the flags of all the arguments are known.)
        }
        for (std::uint32_t i = 0; i < n_nodes; ++i) {
            virt[i] = (ok_virt[i] != 0 && used[i] != 0) ? 1 : 0;
        }
        // Consumers want uniform shapes: a sum of squares all of whose arguments are virtual (or none), a product with at
        // most one virtual factor; and a virtual difference must not read another one.
        bool changed = true;
        while (changed) {
            changed = false;
            const auto is_virt = [&](const operand &o) { return is_uvar(o) && o.idx >= n_eq && virt[o.idx - n_eq] != 0; };
            const auto drop = [&](const operand &o) {
                if (is_virt(o)) {
                    virt[o.idx - n_eq] = 0;
                    changed = true;
                }
            };
            for (std::uint32_t i = 0; i < n_nodes; ++i) {
                const auto &n = p.nodes[i];
                if (spec_of[i] == spec::sum_sq) {
                    const auto nv = std::count_if(n.args.begin(), n.args.end(), is_virt);
                    if (nv != 0 && static_cast<std::size_t>(nv) != n.args.size()) {
                        for (const auto &o : n.args) {
                            drop(o);
                        }
                    }
                } else if (spec_of[i] == spec::prod && n.args.size() == 2u && is_virt(n.args[0]) && is_virt(n.args[1])) {
                    drop(n.args[1]);
                } else if (virt[i] != 0 && (is_virt(n.args[0]) || is_virt(n.args[1]))) {
                    virt[i] = 0;
                    changed = true;
                }
            }
        }
    }
    std::vector<char> hist(n_u, 0);
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        hist[i] = 1;
    }
    for (const auto u : p.ev_u) {
        hist[u] = 1;
    }
    for (std::uint32_t i = 0; i < n_nodes; ++i) {
        const auto &n = p.nodes[i];
        const auto mark_arg = [&](std::size_t a) {
            if (a < n.args.size() && is_uvar(n.args[a])) {
                hist[n.args[a].idx] = 1;
                // (A virtual difference is read through the rows of ITS operands.)
                if (n.args[a].idx >= n_eq && virt[n.args[a].idx - n_eq] != 0) {
                    for (const auto &o2 : p.nodes[n.args[a].idx - n_eq].args) {
                        hist[o2.idx] = 1;
                    }
                }
            }
        };
        const auto self = n_eq + i;
        switch (spec_of[i]) {
            case spec::generic:
                hist[self] = 1;
                for (std::size_t a = 0; a < n.args.size(); ++a) {
                    mark_arg(a);
                }
                for (const auto d : n.deps) {
                    hist[d] = 1;
                }
                break;
            case spec::prod:
                if (n.args.size() == 2u && is_uvar(n.args[0]) && is_uvar(n.args[1])) {
                    mark_arg(0);
                    mark_arg(1);
                }
                break;
            case spec::div:
                if (is_uvar(n.args[1])) {
                    mark_arg(1);
                    hist[self] = 1;
                }
                break;
            case spec::sum_sq:
                for (std::size_t a = 0; a < n.args.size(); ++a) {
                    mark_arg(a);
                }
                break;
            case spec::pow:
            case spec::exp:
            case spec::log:
                mark_arg(0);
                hist[self] = 1;
                break;
            case spec::sin:
            case spec::cos:
                mark_arg(0);
                hist[n.deps[0]] = 1;
                break;
            default:
                break;
        }
    }
    // Fusion of the dependency chain of an order. Every level costs a store - barrier - load round trip through LDS, and an
    // order is a CHAIN of levels (its latency, not its work, bounds the step). Two kinds of u variable need no level of
    // their own:
    //   * followers: a linear function of ONE u variable and numbers / parameters (c * u, -u, u - c, c - u, u / c: the
    //     scalings G m r^-3 and the reactions c (d r^-3) of an N-body system) is computed by the lane(s) which have just
    //     computed that variable, from the value in their register;
    //   * the state-variable recursion x^[k+1] = rhs^[k] / (k + 1) (taylor_compute_sv_diff(), src/taylor_02.cpp:245-287):
    //     written by the lane which computes rhs^[k]; a state variable defined by another state variable (x' = v) follows one
    //     order later, x^[k+2] = v^[k+1] / (k + 2), from the same lane.
    // Same operations on the same operands as the separate nodes: bit for bit the same results.
    std::vector<int> parent(n_nodes, -1);
    std::vector<std::vector<std::uint32_t>> followers(n_nodes), sv_of(n_nodes);
    std::vector<std::vector<std::uint32_t>> chain_of(n_eq); // state variables defined by state variable i
    std::vector<char> sv_fused(n_eq, 0);
    // Measured (profiles/r06_staged_fusion_ab.log, 262 144 systems): followers + 12 % on the outer Solar System (8 -> 5
    // levels), + 0 ... 3 % elsewhere; the fused recursion of the state variables - 2 ... - 12 % on everything but the outer
    // Solar System (+ 1 %): two divisions in sequence on every lane of the last level cost more than the level they save.
    // Default: followers only. (HEYOKA_AMD_TABLE_LDS=3: no fusion, 5: both - A/B switches.)
    const bool fuse = opts.dev.table_lds != 3;
    const bool fuse_sv = opts.dev.table_lds == 5;
    for (std::uint32_t i = 0; i < n_nodes && fuse; ++i) {
        const auto &n = p.nodes[i];
        if (spec_of[i] != spec::prod && spec_of[i] != spec::sub && spec_of[i] != spec::div) {
            continue;
        }
        if (n.args.size() != 2u || is_uvar(n.args[0]) == is_uvar(n.args[1])) {
            continue;
        }
        const auto &ov = is_uvar(n.args[0]) ? n.args[0] : n.args[1];
        if (ov.idx < n_eq || spec_of[ov.idx - n_eq] == spec::generic) {
            continue;
        }
        if (spec_of[i] == spec::div && !is_uvar(n.args[0])) {
            continue; // (c / u is a recurrence, not a linear function of u)
        }
        parent[i] = static_cast<int>(ov.idx - n_eq);
        followers[ov.idx - n_eq].push_back(i);
    }
    for (std::uint32_t i = 0; i < n_eq && fuse_sv; ++i) {
        const auto &d = p.sv_defs[i];
        if (is_uvar(d) && d.idx >= n_eq && spec_of[d.idx - n_eq] != spec::generic) {
            sv_of[d.idx - n_eq].push_back(i);
            sv_fused[i] = 1;
        }
    }
    for (std::uint32_t i = 0; i < n_eq && fuse_sv; ++i) {
        const auto &d = p.sv_defs[i];
        // (One link: x' = v with v' a node. Longer chains of state variables keep the separate recursion.)
        if (is_uvar(d) && d.idx < n_eq && sv_fused[d.idx] != 0 && !(is_uvar(p.sv_defs[d.idx]) && p.sv_defs[d.idx].idx < n_eq)) {
            chain_of[d.idx].push_back(i);
            sv_fused[i] = 1;
        }
    }
    // Levels again: a follower sits at the level of the variable it follows.
    n_levels = 0;
    for (std::uint32_t i = 0; i < n_nodes; ++i) {
        std::uint32_t l = 0;
        if (parent[i] >= 0) {
            l = level[static_cast<std::uint32_t>(parent[i])];
        } else {
            for (const auto &o : p.nodes[i].args) {
                if (is_uvar(o) && o.idx >= n_eq) {
                    if (virt[o.idx - n_eq] != 0) {
                        // (Read through the rows of its operands: no level of its own.)
                        for (const auto &o2 : p.nodes[o.idx - n_eq].args) {
                            if (o2.idx >= n_eq) {
                                l = std::max(l, level[o2.idx - n_eq] + 1u);
                            }
                        }
                    } else {
                        l = std::max(l, level[o.idx - n_eq] + 1u);
                    }
                }
            }
        }
        level[i] = l;
        n_levels = std::max(n_levels, l + 1u);
    }
    // Shape of what hangs off a node: part of the key of its group (the lanes of a group run one instruction stream).
    std::function<std::string(std::uint32_t)> tail_sig = [&](std::uint32_t i) {
        std::string sg;
        for (const auto f : followers[i]) {
            sg += "{" + key_of[f] + (hist[n_eq + f] != 0 ? "H" : "S") + tail_sig(f) + "}";
        }
        if (!sv_of[i].empty()) {
            sg += "|sv" + std::to_string(sv_of[i].size());
            for (const auto v : sv_of[i]) {
                sg += "c" + std::to_string(chain_of[v].size());
            }
        }
        return sg;
    };

    // Layout: rows of the history variables (state variables first: row i = state variable i), the dummy row, the slab.
    std::vector<std::uint32_t> base(n_u, 0u);
    std::uint32_t n_hist = 0, n_slab = 0;
    const auto is_virt_u = [&](std::uint32_t u) { return u >= n_eq && virt[u - n_eq] != 0; };
    for (std::uint32_t u = 0; u < n_u; ++u) {
        if (hist[u] != 0 && !is_virt_u(u)) {
            base[u] = (n_hist++) * P;
        }
    }
    const std::uint32_t dummy_row = n_hist * P;
    for (std::uint32_t u = 0; u < n_u; ++u) {
        if (hist[u] == 0 && !is_virt_u(u)) {
            base[u] = (n_hist + 1u) * P + (n_slab++);
        }
    }
    const auto tape_doubles = static_cast<std::uint64_t>(n_hist + 1u) * P + n_slab + 1u;
    const auto tape_bytes = tape_doubles * sizeof(double);
    if (tape_bytes + lds_reserve > lds_per_cu) {
        why_not = "the tape of one system (" + std::to_string(tape_bytes) + " B) does not fit in the LDS of a CU";
        return ret;
    }
    if (tape_doubles >= (1ull << 31)) {
        why_not = "tape offsets beyond 31 bits";
        return ret;
    }
    // Systems per CU by LDS, and the lanes of a workgroup. The tapes limit a CU to a few systems (7 for the outer Solar
    // System); the lanes of a system are spread over the nodes of a group AND - where a group is small - over the terms of
    // their convolutions (below).
    const auto per_cu = std::min<std::uint64_t>(lds_per_cu / (tape_bytes + lds_reserve), 32u);
    // Measured (profiles/r06_staged_wps.log, 262 144 systems): ONE wavefront per system wins on every decomposition - 7.2e7
    // against 5.5e7 (two) and 3.5e7 (four wavefronts) on the outer Solar System, 2.5e7 / 1.9e7 / 5.9e6 on 511 u variables:
    // a level then ends with a wave-level synchronisation (LDS operations of a wavefront complete in order: no s_barrier),
    // and the 5 ... 10 systems of a CU hide each other's latency. Four wavefronts only where ONE tape fills the LDS of a CU.
    std::uint32_t wps = per_cu >= 2u ? 1u : 4u;
    if (const char *e = std::getenv("HEYOKA_AMD_STAGED_WPS")) {
        // (Experiment switch: wavefronts per system.)
        const auto w = static_cast<std::uint32_t>(std::atoi(e));
        if (w == 1u || w == 2u || w == 4u) {
            wps = w;
        }
    }
    const std::uint32_t LANES = 64u * wps;

    std::vector<group> groups;
    {
        std::map<std::pair<std::uint32_t, std::string>, std::size_t> gidx;
        for (std::uint32_t i = 0; i < n_nodes; ++i) {
            if (parent[i] >= 0 || virt[i] != 0) {
                continue; // (computed by the lanes of the variable it follows / never stored)
            }
            // (Whether the result and the arguments sit in a row or in a slab cell is part of the shape of the code.)
            std::string key = key_of[i];
            if (spec_of[i] != spec::generic) {
                key += hist[n_eq + i] != 0 ? ":H" : ":S";
                for (const auto &o : p.nodes[i].args) {
                    key += !is_uvar(o) ? '-' : (is_virt_u(o.idx) ? 'V' : (hist[o.idx] != 0 ? 'h' : 's'));
                }
                key += tail_sig(i);
            }
            const auto s = spec_of[i];
            const auto kk = std::make_pair(level[i], key);
            auto it = gidx.find(kk);
            if (it == gidx.end()) {
                it = gidx.emplace(kk, groups.size()).first;
                groups.emplace_back();
                groups.back().level = level[i];
                groups.back().sp = s;
                groups.back().key = key;
            }
            groups[it->second].nodes.push_back(i);
        }
        std::stable_sort(groups.begin(), groups.end(), [](const group &a, const group &b) { return a.level < b.level; });
    }

    // ---- code of the groups ----
    lane_tables lt;
    lt.lanes = LANES;
    std::ostringstream o0, ok; // order 0 / order k (runtime k >= 1)
    const auto row_of = [&](std::uint32_t u) { return base[u]; };
    std::uint32_t n_rounds_total = 0, n_generic = 0, n_split = 0;
    std::map<std::string, bool> reg_hist; // register of a row -> whole row (true) or slab cell (false)
    const char *sync = wps == 1u ? "HY_WSYNC();\n" : "__syncthreads();\n";

    for (std::size_t g = 0; g < groups.size(); ++g) {
        const auto &grp = groups[g];
        const auto &n0 = p.nodes[grp.nodes[0]];
        const auto ng = static_cast<std::uint32_t>(grp.nodes.size());
        const auto nargs = n0.args.size();
        const auto v0 = nargs > 0u && is_uvar(n0.args[0]);
        const auto v1 = nargs > 1u && is_uvar(n0.args[1]);
        // Rules with a convolution (a loop over the lower orders), and how their work is laid out on the lanes: `nu` units
        // per node (the arguments of a sum of squares are convolutions of their own), `sp` lanes per unit.
        bool conv = false;
        switch (grp.sp) {
            case spec::prod:
                conv = v0 && v1;
                break;
            case spec::div:
                conv = v1;
                break;
            case spec::sum_sq:
            case spec::pow:
            case spec::sin:
            case spec::cos:
            case spec::exp:
            case spec::log:
                conv = true;
                break;
            default:
                break;
        }
        const std::uint32_t nu = (grp.sp == spec::sum_sq && !strict) ? static_cast<std::uint32_t>(nargs) : 1u;
        std::uint32_t sp = 1;
        if (conv && !strict) {
            // As many lanes per unit (1, 2, 4) as keep the group within one round of the workgroup.
            while (sp < 4u && static_cast<std::uint64_t>(ng) * nu * (sp * 2u) <= LANES && nu * sp * 2u <= 64u) {
                sp *= 2u;
            }
        }
        n_split += sp > 1u ? 1u : 0u;
        const std::uint32_t lpn = nu * sp;                   // lanes per node
        const std::uint32_t npw = 64u / lpn;                 // nodes per wavefront (a node never straddles two)
        const std::uint32_t npr = npw * wps;                 // nodes per round
        const auto rounds = (ng + npr - 1u) / npr;
        if (g > 0u && groups[g - 1u].level != grp.level) {
            o0 << sync;
            ok << sync;
        }
        o0 << "// level " << grp.level << ", " << grp.key << " x " << ng << "\n";
        ok << "// level " << grp.level << ", " << grp.key << " x " << ng << (sp > 1u || nu > 1u ? " (" + std::to_string(lpn) + " lanes per node)" : "") << "\n";
        for (std::uint32_t r = 0; r < rounds; ++r) {
            ++n_rounds_total;
            // Node, unit and role of every lane in this round. Idle lanes replicate the first node of the round; only the
            // first lane of a node (the leader) owns its tape row, all the others write the dummy row.
            std::vector<std::uint32_t> node_of(LANES), unit_of(LANES);
            std::vector<char> leader(LANES, 0);
            for (std::uint32_t l = 0; l < LANES; ++l) {
                const auto w = l / 64u, loc = l % 64u;
                const auto slot = loc / lpn, rem = loc % lpn;
                const auto j = r * npr + w * npw + slot;
                const bool live = slot < npw && j < ng;
                node_of[l] = grp.nodes[live ? j : r * npr];
                unit_of[l] = rem / sp;
                leader[l] = (live && rem == 0u) ? 1 : 0;
            }
            const auto urow = [&](auto &&f) {
                std::vector<std::uint32_t> v(LANES);
                for (std::uint32_t l = 0; l < LANES; ++l) {
                    v[l] = f(l);
                }
                return lt.add_u(std::move(v));
            };
            const std::string O
                = urow([&](std::uint32_t l) { return leader[l] != 0 ? row_of(n_eq + node_of[l]) : dummy_row; });
            reg_hist[O] = grp.sp == spec::generic || hist[n_eq + grp.nodes[0]] != 0;
            // (Recurrences on the node's own lower orders are READ by all the lanes of the node.)
            const auto own_row = [&]() {
                const auto nm = urow([&](std::uint32_t l) { return row_of(n_eq + node_of[l]); });
                reg_hist[nm] = true;
                return nm;
            };
            if (grp.sp == spec::generic) {
                ++n_generic;
                const std::string NI = urow([&](std::uint32_t l) { return node_of[l]; });
                o0 << "HY_T(" << O << ", 0u) = hy_node_value(c, " << NI << ", 0u);\n";
                ok << "HY_T(" << O << ", k) = hy_node_value(c, " << NI << ", k);\n";
                continue;
            }
            // Arguments: tape rows of the u variables; numbers as literals when every node of the group has the same
            // value, per-lane registers otherwise; parameters as per-lane values loaded with the system.
            // (A2[a]: the row of the subtrahend when argument a is a virtual difference - A[a] then holds the minuend's.)
            std::vector<std::string> A(nargs), A2(nargs), C(nargs);
            const auto arg_rows = [&](const std::function<const operand &(std::uint32_t)> &arg_of, std::string &r1, std::string &r2) {
                if (is_virt_u(arg_of(0).idx)) {
                    r1 = urow([&](std::uint32_t l) { return row_of(p.nodes[arg_of(l).idx - n_eq].args[0].idx); });
                    r2 = urow([&](std::uint32_t l) { return row_of(p.nodes[arg_of(l).idx - n_eq].args[1].idx); });
                    reg_hist[r1] = true;
                    reg_hist[r2] = true;
                } else {
                    r1 = urow([&](std::uint32_t l) { return row_of(arg_of(l).idx); });
                    reg_hist[r1] = hist[arg_of(0).idx] != 0;
                }
            };
            for (std::size_t a = 0; a < nargs; ++a) {
                const auto &o = n0.args[a];
                if (is_uvar(o)) {
                    if (nu > 1u) {
                        // (Sum of squares over units: every lane holds the row of ITS argument, in A[0].)
                        continue;
                    }
                    arg_rows([&, a](std::uint32_t l) -> const operand & { return p.nodes[node_of[l]].args[a]; }, A[a], A2[a]);
                } else if (o.type == operand::kind::par) {
                    std::vector<std::uint32_t> v(LANES);
                    for (std::uint32_t l = 0; l < LANES; ++l) {
                        v[l] = p.nodes[node_of[l]].args[a].idx;
                    }
                    C[a] = lt.add_p(std::move(v));
                } else {
                    bool same = true;
                    for (const auto i : grp.nodes) {
                        const auto x = p.nodes[i].args[a].value;
                        same = same && (x == o.value || (x != x && o.value != o.value)) && std::signbit(x) == std::signbit(o.value);
                    }
                    if (same) {
                        C[a] = fp_literal(o.value);
                    } else {
                        std::vector<double> v(LANES);
                        for (std::uint32_t l = 0; l < LANES; ++l) {
                            v[l] = p.nodes[node_of[l]].args[a].value;
                        }
                        C[a] = lt.add_d(std::move(v));
                    }
                }
            }
            if (nu > 1u) {
                arg_rows([&](std::uint32_t l) -> const operand & { return p.nodes[node_of[l]].args[unit_of[l]]; }, A[0], A2[0]);
                reg_hist[A[0]] = true;
            }
            std::string D;
            if (grp.sp == spec::sin || grp.sp == spec::cos) {
                D = urow([&](std::uint32_t l) { return row_of(p.nodes[node_of[l]].deps[0]); });
                reg_hist[D] = true;
            }
            // Order kk of the u variable whose row / cell register is `row`.
            const auto T = [&](const std::string &row, const std::string &kk) {
                return reg_hist.at(row) ? "HY_T(" + row + ", " + kk + ")" : "hy_lds_tape[" + row + "]";
            };
            // Loop header of a convolution whose index j runs from j0 to jend (inclusive if incl): this lane's share.
            const auto jloop = [&](const std::string &j0, const std::string &cond) {
                std::ostringstream h;
                const bool full = opts.dev.table_lds == 7 || opts.dev.table_lds == 8;
                if (sp == 1u) {
                    h << (full ? "#pragma unroll\n" : "#pragma unroll 4\n") << "for (unsigned j = " << j0 << "; " << cond << "; ++j)";
                } else {
                    h << (full ? "#pragma unroll\n" : "#pragma unroll 2\n") << "for (unsigned j = " << j0 << " + (lane & " << (sp - 1u)
                      << "u); " << cond << "; j += " << sp << "u)";
                }
                return h.str();
            };
            // Sum of the partial sums of the lanes of a unit (every lane ends up with the total).
            const auto reduce = [&](const std::string &v) {
                std::ostringstream h;
                if (sp >= 2u) {
                    h << v << " = " << v << " + hy_dpp<0xB1>(" << v << ");\n";
                }
                if (sp >= 4u) {
                    h << v << " = " << v << " + hy_dpp<0x4E>(" << v << ");\n";
                }
                return h.str();
            };
            std::ostringstream r0, rk;
            switch (grp.sp) {
                case spec::sum: {
                    // hy_diff_sum(): numbers / parameters contribute at order 0 only; pairwise over all the arguments.
                    std::vector<std::string> t0, tk;
                    for (std::size_t a = 0; a < nargs; ++a) {
                        t0.push_back(is_uvar(n0.args[a]) ? T(A[a], "0u") : C[a]);
                        tk.push_back(is_uvar(n0.args[a]) ? T(A[a], "k") : std::string("0.0"));
                    }
                    r0 << T(O, "0u") << " = " << pairwise(t0) << ";\n";
                    rk << T(O, "k") << " = " << pairwise(tk) << ";\n";
                    break;
                }
                case spec::sub:
                    if (v0 && v1) {
                        r0 << T(O, "0u") << " = " << T(A[0], "0u") << " - " << T(A[1], "0u") << ";\n";
                        rk << T(O, "k") << " = " << T(A[0], "k") << " - " << T(A[1], "k") << ";\n";
                    } else if (v0) {
                        r0 << T(O, "0u") << " = " << T(A[0], "0u") << " - " << C[1] << ";\n";
                        rk << T(O, "k") << " = " << T(A[0], "k") << ";\n";
                    } else if (v1) {
                        r0 << T(O, "0u") << " = " << C[0] << " - " << T(A[1], "0u") << ";\n";
                        rk << T(O, "k") << " = -" << T(A[1], "k") << ";\n";
                    } else {
                        r0 << T(O, "0u") << " = " << C[0] << " - " << C[1] << ";\n";
                        rk << T(O, "k") << " = 0.0;\n";
                    }
                    break;
                case spec::prod: {
                    const bool neg = n0.args[0].type == operand::kind::num && n0.args[0].value == -1.;
                    if (v0 && v1) {
                        // (A virtual factor is read as the difference of the rows of its operands.)
                        const auto op0 = [&](const std::string &ptr, const std::string &idx, std::size_t a) {
                            return A2[a].empty() ? ptr + "[" + idx + "]" : "(" + ptr + "[" + idx + "] - " + ptr + "2[" + idx + "])";
                        };
                        const auto val0 = [&](std::size_t a) {
                            return A2[a].empty() ? T(A[a], "0u") : "(" + T(A[a], "0u") + " - " + T(A2[a], "0u") + ")";
                        };
                        r0 << T(O, "0u") << " = " << val0(0) << " * " << val0(1) << ";\n";
                        rk << "{\nconst double *pa = hy_lds_tape + " << A[0] << " + k, *pb = hy_lds_tape + " << A[1] << ";\n";
                        if (!A2[0].empty()) {
                            rk << "const double *pa2 = hy_lds_tape + " << A2[0] << " + k;\n";
                        }
                        if (!A2[1].empty()) {
                            rk << "const double *pb2 = hy_lds_tape + " << A2[1] << ";\n";
                        }
                        rk << "double acc = 0.0;\n" << jloop("0u", "j <= k") << " acc += " << op0("pa", "-(int)j", 0) << " * " << op0("pb", "j", 1)
                           << ";\n" << reduce("acc") << T(O, "k") << " = acc;\n}\n";
                    } else if (!v0 && !v1) {
                        r0 << T(O, "0u") << " = " << (neg ? "-" + C[1] : C[0] + " * " + C[1]) << ";\n";
                        rk << T(O, "k") << " = 0.0;\n";
                    } else {
                        const auto av = v0 ? 0u : 1u, an = v0 ? 1u : 0u;
                        const std::string f = (neg && an == 0u) ? std::string("-") : (C[an] + " * ");
                        r0 << T(O, "0u") << " = " << f << T(A[av], "0u") << ";\n";
                        rk << T(O, "k") << " = " << f << T(A[av], "k") << ";\n";
                    }
                    break;
                }
                case spec::div:
                    if (v1) {
                        const auto U = own_row();
                        r0 << T(O, "0u") << " = " << (v0 ? T(A[0], "0u") : C[0]) << " / " << T(A[1], "0u") << ";\n";
                        rk << "{\nconst double *pu = hy_lds_tape + " << U << " + k, *pd = hy_lds_tape + " << A[1]
                           << ";\ndouble acc = 0.0;\n" << jloop("1u", "j <= k") << " acc += pu[-(int)j] * pd[j];\n" << reduce("acc")
                           << T(O, "k") << " = " << (v0 ? "(" + T(A[0], "k") + " - acc)" : std::string("(-acc)")) << " / pd[0];\n}\n";
                    } else if (v0) {
                        r0 << T(O, "0u") << " = " << T(A[0], "0u") << " / " << C[1] << ";\n";
                        rk << T(O, "k") << " = " << T(A[0], "k") << " / " << C[1] << ";\n";
                    } else {
                        r0 << T(O, "0u") << " = " << C[0] << " / " << C[1] << ";\n";
                        rk << T(O, "k") << " = 0.0;\n";
                    }
                    break;
                case spec::sum_sq: {
                    // hy_diff_sum_sq(): per-argument running sums, pairwise sum over the arguments, doubled at odd orders.
                    rk << "{\nconst unsigned odd = k & 1u, jn = odd ? (k + 1u) / 2u : k / 2u;\n";
                    if (nu == 1u) {
                        std::vector<std::string> t0, tk;
                        for (std::size_t a = 0; a < nargs; ++a) {
                            const bool va = !A2[a].empty();
                            const auto x0 = va ? "(" + T(A[a], "0u") + " - " + T(A2[a], "0u") + ")" : T(A[a], "0u");
                            t0.push_back("(" + x0 + " * " + x0 + ")");
                            const auto s_ = std::to_string(a);
                            const auto op = [&](const std::string &idx) { return va ? "(pa[" + idx + "] - pa2[" + idx + "])" : "pa[" + idx + "]"; };
                            rk << "double t" << s_ << ";\n{\nconst double *pa = hy_lds_tape + " << A[a] << ";\n";
                            if (va) {
                                rk << "const double *pa2 = hy_lds_tape + " << A2[a] << ";\n";
                            }
                            rk << "double acc = 0.0;\n" << jloop("0u", "j < jn") << " acc += " << op("k - j") << " * " << op("j") << ";\n"
                               << reduce("acc") << "const double hv = " << op("k / 2u") << ";\nt" << s_
                               << " = odd ? acc : (acc + acc) + hv * hv;\n}\n";
                            tk.push_back("t" + s_);
                        }
                        r0 << T(O, "0u") << " = " << pairwise(t0) << ";\n";
                        rk << "const double tot = " << pairwise(tk) << ";\n" << T(O, "k") << " = odd ? tot + tot : tot;\n}\n";
                    } else {
                        // One unit of `sp` lanes per argument: the terms of the arguments are collected by wave shuffles
                        // from the first lane of every unit (lane L0 + a * sp), then added pairwise like above.
                        const std::string L0 = urow([&](std::uint32_t l) { return (l % 64u) - ((l % 64u) % lpn); });
                        const bool va = !A2[0].empty();
                        const auto op = [&](const std::string &idx) { return va ? "(pa[" + idx + "] - pa2[" + idx + "])" : "pa[" + idx + "]"; };
                        r0 << "{\nconst double x0 = " << (va ? T(A[0], "0u") + " - " + T(A2[0], "0u") : T(A[0], "0u"))
                           << ";\nconst double q0 = x0 * x0;\n";
                        rk << "const double *pa = hy_lds_tape + " << A[0] << ";\n";
                        if (va) {
                            rk << "const double *pa2 = hy_lds_tape + " << A2[0] << ";\n";
                        }
                        rk << "double acc = 0.0;\n" << jloop("0u", "j < jn") << " acc += " << op("k - j") << " * " << op("j") << ";\n"
                           << reduce("acc") << "const double hv = " << op("k / 2u") << ";\nconst double tu = odd ? acc : (acc + acc) + hv * hv;\n";
                        std::vector<std::string> t0, tk;
                        for (std::size_t a = 0; a < nargs; ++a) {
                            const auto s_ = std::to_string(a);
                            r0 << "const double q0_" << s_ << " = __shfl(q0, (int)(" << L0 << " + " << a * sp << "u), 64);\n";
                            rk << "const double t" << s_ << " = __shfl(tu, (int)(" << L0 << " + " << a * sp << "u), 64);\n";
                            t0.push_back("q0_" + s_);
                            tk.push_back("t" + s_);
                        }
                        r0 << T(O, "0u") << " = " << pairwise(t0) << ";\n}\n";
                        rk << "const double tot = " << pairwise(tk) << ";\n" << T(O, "k") << " = odd ? tot + tot : tot;\n}\n";
                    }
                    break;
                }
                case spec::pow: {
                    const auto ex = n0.args[1].value;
                    const auto U = own_row();
                    r0 << T(O, "0u") << " = hy_pow_eval(" << T(A[0], "0u") << ", " << fp_literal(ex) << ");\n";
                    if (ex == 0.5) {
                        // sqrt (src/math/pow.cpp:432-474).
                        rk << "{\nconst double *pu = hy_lds_tape + " << U << ";\nconst double a_0 = pu[0];\ndouble fac = "
                           << T(A[0], "k") << ", acc = 0.0;\nconst unsigned jmax = (k & 1u) ? (k - 1u) / 2u : (k - 2u) / 2u;\n"
                           << jloop("1u", "j <= jmax") << " acc += pu[k - j] * pu[j];\n" << reduce("acc")
                           << "if ((k & 1u) == 0u) { const double hv = pu[k / 2u]; fac = fac - hv * hv; }\n"
                           << "if (jmax >= 1u) fac = fac - (acc + acc);\n"
                           << T(O, "k") << " = fac / (a_0 + a_0);\n}\n";
                    } else if (ex == 2.) {
                        // square (src/math/pow.cpp:395-430).
                        rk << "{\nconst double *pa = hy_lds_tape + " << A[0]
                           << ";\nconst unsigned odd = k & 1u, jn = odd ? (k + 1u) / 2u : k / 2u;\ndouble acc = 0.0;\n"
                           << jloop("0u", "j < jn") << " acc += pa[k - j] * pa[j];\n" << reduce("acc")
                           << "const double hv = pa[k / 2u];\n"
                           << T(O, "k") << " = odd ? acc + acc : (acc + acc) + hv * hv;\n}\n";
                    } else {
                        rk << "{\nconst double *pb = hy_lds_tape + " << A[0] << " + k, *pu = hy_lds_tape + " << U
                           << ";\nconst double ex = " << fp_literal(ex) << ", kex = (double)k * ex;\ndouble acc = 0.0;\n"
                           << jloop("0u", "j < k") << " {\nconst double sf = kex - (double)j * (ex + 1.0);\n"
                           << "acc += sf * (pb[-(int)j] * pu[j]);\n}\n" << reduce("acc")
                           << T(O, "k") << " = acc / ((double)k * " << T(A[0], "0u") << ");\n}\n";
                    }
                    break;
                }
                case spec::sin:
                case spec::cos: {
                    const bool is_sin = grp.sp == spec::sin;
                    r0 << T(O, "0u") << " = " << (is_sin ? "hy_sin(" : "hy_cos(") << T(A[0], "0u") << ");\n";
                    rk << "{\nconst double *pd = hy_lds_tape + " << D << " + k, *pb = hy_lds_tape + " << A[0]
                       << ";\ndouble acc = 0.0;\n" << jloop("1u", "j <= k") << " acc += (double)j * (pd[-(int)j] * pb[j]);\n"
                       << reduce("acc") << T(O, "k") << " = acc / " << (is_sin ? "(double)k" : "-(double)k") << ";\n}\n";
                    break;
                }
                case spec::exp: {
                    const auto U = own_row();
                    r0 << T(O, "0u") << " = exp(" << T(A[0], "0u") << ");\n";
                    rk << "{\nconst double *pu = hy_lds_tape + " << U << " + k, *pb = hy_lds_tape + " << A[0]
                       << ";\ndouble acc = 0.0;\n" << jloop("1u", "j <= k") << " acc += (double)j * (pu[-(int)j] * pb[j]);\n"
                       << reduce("acc") << T(O, "k") << " = acc / (double)k;\n}\n";
                    break;
                }
                case spec::log: {
                    const auto U = own_row();
                    r0 << T(O, "0u") << " = log(" << T(A[0], "0u") << ");\n";
                    rk << "{\nconst double *pb = hy_lds_tape + " << A[0] << " + k, *pu = hy_lds_tape + " << U
                       << ";\ndouble ret = (double)k * pb[0];\ndouble acc = 0.0;\n" << jloop("1u", "j < k")
                       << " acc += (double)j * (pb[-(int)j] * pu[j]);\n" << reduce("acc") << "if (k > 1u) ret = ret - acc;\n"
                       << T(O, "k") << " = ret / ((double)k * " << T(A[0], "0u") << ");\n}\n";
                    break;
                }
                case spec::time:
                    r0 << T(O, "0u") << " = t_hi;\n";
                    rk << T(O, "k") << " = (k == 1u) ? 1.0 : 0.0;\n";
                    break;
                case spec::number:
                    r0 << T(O, "0u") << " = " << C[0] << ";\n";
                    rk << T(O, "k") << " = 0.0;\n";
                    break;
                default:
                    break;
            }
            // The value of the node stays in a register (hy_v): its row / cell, then what follows it - linear functions of it
            // and the recursion of the state variables it defines - without another trip through LDS (see `followers`).
            const auto finish = [&](std::ostringstream &dst, const std::string &body, const std::string &kk) {
                const auto tgt = T(O, kk) + " = ";
                std::string b = body;
                for (std::size_t pos = b.find(tgt); pos != std::string::npos; pos = b.find(tgt, pos)) {
                    b.replace(pos, tgt.size(), "hy_v = ");
                    pos += 7u;
                }
                dst << "{\ndouble hy_v;\n" << b << tgt << "hy_v;\n";
                std::uint32_t depth = 0;
                // f_of(l): the node whose value sits in `var` on lane l.
                std::function<void(const std::function<std::uint32_t(std::uint32_t)> &, const std::string &)> tail
                    = [&](const std::function<std::uint32_t(std::uint32_t)> &f_of, const std::string &var) {
                          const auto rep = f_of(0);
                          // Followers.
                          for (std::size_t fi = 0; fi < followers[rep].size(); ++fi) {
                              const auto sel = [&, fi](std::uint32_t l) { return followers[f_of(l)][fi]; };
                              const auto &nf = p.nodes[sel(0)];
                              const auto an = is_uvar(nf.args[0]) ? 1u : 0u; // position of the number / parameter
                              std::string cst;
                              if (nf.args[an].type == operand::kind::par) {
                                  std::vector<std::uint32_t> v(LANES);
                                  for (std::uint32_t l = 0; l < LANES; ++l) {
                                      v[l] = p.nodes[sel(l)].args[an].idx;
                                  }
                                  cst = lt.add_p(std::move(v));
                              } else {
                                  bool same = true;
                                  for (std::uint32_t l = 0; l < LANES; ++l) {
                                      const auto x = p.nodes[sel(l)].args[an].value, y = nf.args[an].value;
                                      same = same && (x == y || (x != x && y != y)) && std::signbit(x) == std::signbit(y);
                                  }
                                  if (same) {
                                      cst = fp_literal(nf.args[an].value);
                                  } else {
                                      std::vector<double> v(LANES);
                                      for (std::uint32_t l = 0; l < LANES; ++l) {
                                          v[l] = p.nodes[sel(l)].args[an].value;
                                      }
                                      cst = lt.add_d(std::move(v));
                                  }
                              }
                              std::string ex;
                              const bool k0 = kk == "0u";
                              switch (spec_of[sel(0)]) {
                                  case spec::prod:
                                      // (hy_diff_prod(): a leading number -1 negates.)
                                      ex = (an == 0u && nf.args[0].type == operand::kind::num && nf.args[0].value == -1.) ? "-" + var
                                                                                                                           : cst + " * " + var;
                                      break;
                                  case spec::sub:
                                      ex = an == 1u ? (k0 ? var + " - " + cst : var) : (k0 ? cst + " - " + var : "-" + var);
                                      break;
                                  default: // div: u / c
                                      ex = var + " / " + cst;
                                      break;
                              }
                              const auto nv = "hy_w" + std::to_string(depth++);
                              const auto OF = urow([&](std::uint32_t l) { return leader[l] != 0 ? row_of(n_eq + sel(l)) : dummy_row; });
                              reg_hist[OF] = hist[n_eq + sel(0)] != 0;
                              dst << "const double " << nv << " = " << ex << ";\n" << T(OF, kk) << " = " << nv << ";\n";
                              tail(sel, nv);
                          }
                          // State variables defined by this u variable: x^[k+1] = u^[k] / (k + 1), and the state variables
                          // defined by THOSE one order further.
                          for (std::size_t si = 0; si < sv_of[rep].size(); ++si) {
                              const auto sv_sel = [&, si](std::uint32_t l) { return sv_of[f_of(l)][si]; };
                              const auto SV = urow([&](std::uint32_t l) { return leader[l] != 0 ? row_of(sv_sel(l)) : dummy_row; });
                              const bool k0 = kk == "0u";
                              const auto sv1 = "hy_x" + std::to_string(depth++);
                              dst << "const double " << sv1 << " = " << var << " / " << (k0 ? "1.0" : "(double)(k + 1u)") << ";\n";
                              dst << (k0 ? "" : "if (k + 1u <= HY_ORDER) ") << "hy_lds_tape[" << SV << " + " << (k0 ? "1u" : "k + 1u") << "] = " << sv1
                                  << ";\n";
                              for (std::size_t ci = 0; ci < chain_of[sv_sel(0)].size(); ++ci) {
                                  const auto CH = urow([&](std::uint32_t l) {
                                      return leader[l] != 0 ? row_of(chain_of[sv_sel(l)][ci]) : dummy_row;
                                  });
                                  if (k0) {
                                      // x^[1] = v^[0] / 1 (the state), x^[2] = v^[1] / 2.
                                      dst << "hy_lds_tape[" << CH << " + 1u] = hy_lds_tape[" << SV << "] / 1.0;\n";
                                      dst << "hy_lds_tape[" << CH << " + 2u] = " << sv1 << " / 2.0;\n";
                                  } else {
                                      dst << "if (k + 2u <= HY_ORDER) hy_lds_tape[" << CH << " + k + 2u] = " << sv1 << " / (double)(k + 2u);\n";
                                  }
                              }
                          }
                      };
                tail([&](std::uint32_t l) { return node_of[l]; }, "hy_v");
                dst << "}\n";
            };
            finish(o0, r0.str(), "0u");
            finish(ok, rk.str(), "k");
        }
    }
    o0 << sync;
    ok << sync;

    // ---- state variables: recursion x^[k] = rhs^[k-1] / k (taylor_compute_sv_diff(), src/taylor_02.cpp:245-287) ----
    // (Only the state variables whose recursion is not fused into the lanes of their right-hand sides - see `followers`.)
    std::ostringstream svk;
    std::vector<std::uint32_t> sv_rest;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        if (sv_fused[i] == 0) {
            sv_rest.push_back(i);
        }
    }
    const auto n_rest = static_cast<std::uint32_t>(sv_rest.size());
    const auto sv_rounds = (n_rest + LANES - 1u) / LANES;
    bool sv_any_par = false;
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        std::vector<std::uint32_t> own(LANES), def(LANES), isv(LANES), pidx(LANES, 0u), dstr(LANES, 1u);
        std::vector<double> cv(LANES, 0.);
        bool any_var = false, any_num = false, any_par = false;
        for (std::uint32_t l = 0; l < LANES; ++l) {
            const auto i = r * LANES + l;
            const auto ii = sv_rest[i < n_rest ? i : r * LANES];
            const auto &d = p.sv_defs[ii];
            own[l] = i < n_rest ? row_of(ii) : dummy_row;
            isv[l] = is_uvar(d) ? 1u : 0u;
            def[l] = is_uvar(d) ? row_of(d.idx) : dummy_row;
            if (is_uvar(d)) {
                any_var = true;
            } else if (d.type == operand::kind::par) {
                any_par = true;
                pidx[l] = d.idx;
                isv[l] = 2u;
            } else {
                any_num = true;
                cv[l] = d.value;
            }
        }
        const auto O = lt.add_u(own), Dn = lt.add_u(def);
        // (The defining u variable may sit in a row or in a slab cell, lane by lane: per-lane stride 1 / 0.)
        bool any_slab = false;
        for (std::uint32_t l = 0; l < LANES; ++l) {
            const auto i = r * LANES + l;
            const auto &d = p.sv_defs[sv_rest[i < n_rest ? i : r * LANES]];
            dstr[l] = (is_uvar(d) && hist[d.idx] == 0) ? 0u : 1u;
            any_slab = any_slab || dstr[l] == 0u;
        }
        const std::string dval = any_slab ? "hy_lds_tape[" + Dn + " + (k - 1u) * " + lt.add_u(dstr) + "]" : "HY_T(" + Dn + ", k - 1u)";
        std::string val;
        if (any_var && !any_num && !any_par) {
            val = dval + " / (double)k";
        } else {
            const auto F = lt.add_u(isv);
            std::string cst = any_num ? lt.add_d(cv) : std::string("0.0");
            if (any_par) {
                sv_any_par = true;
                const auto Pn = lt.add_p(pidx);
                cst = "((" + F + " == 2u) ? " + Pn + " : " + cst + ")";
            }
            val = "(" + F + " == 1u) ? " + dval + " / (double)k : ((k == 1u) ? " + cst + " : 0.0)";
        }
        svk << "HY_T(" << O << ", k) = " << val << ";\n";
    }
    if (n_rest != 0u) {
        svk << sync;
    }
    (void)sv_any_par;

    // ---- module text ----
    std::ostringstream src;
    src << emit_detail::prelude << emit_detail::rules_source(p);
    emit_detail::emit_dout(src, p, opts);
    src << emit_detail::wsync_macro;
    {
        std::ostringstream defs;
        defs << "#define HY_STAGED 1\n#define HY_P " << P << "u\n#define HY_TAPE_DOUBLES " << tape_doubles << "u\n#define HY_LANES " << LANES << "u\n#define HY_WPS " << wps
             << "u\n#define HY_T(off, kk) hy_lds_tape[(off) + (kk)]\n#define HY_DUMMY " << dummy_row << "u\n";
        defs << "__device__ const unsigned hy_row_of[] = {";
        for (std::uint32_t u = 0; u < n_u; ++u) {
            defs << base[u] << ",";
        }
        defs << "0};\n";
        table_detail::emit_tables_and_rules(src, p, opts, defs.str(), "1u");
    }
    // Per-lane tables: [row][lane].
    const auto put_rows = [&](const char *name, const char *type, const auto &rows, auto &&fmt) {
        src << "__device__ const " << type << " " << name << "[] = {";
        for (const auto &v : rows) {
            for (const auto &x : v) {
                src << fmt(x) << ",";
            }
            src << "\n";
        }
        src << fmt(typename std::decay_t<decltype(rows)>::value_type::value_type{}) << "};\n";
    };
    put_rows("hy_st_u", "unsigned", lt.urows, [](std::uint32_t x) { return std::to_string(x); });
    put_rows("hy_st_d", "double", lt.drows, [](double x) { return fp_literal(x); });
    put_rows("hy_st_p", "unsigned", lt.prows, [](std::uint32_t x) { return std::to_string(x); });

    src << R"HIP(
#if HY_WPS > 1u
__shared__ double hy_red[3u * HY_WPS];
__shared__ u64 hy_sh_base;
__shared__ int hy_sh_flag;
#endif
__device__ __forceinline__ double hy_wave_max(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v = hy_max(v, __shfl_xor(v, m, 64));
    return v;
}
// Order k of state variable i (row i of the tape).
__device__ __forceinline__ double &hy_sv(unsigned i, unsigned k)
{
    return hy_lds_tape[i * HY_P + k];
}
// Exchange inside a quad of lanes (DPP quad_perm on the two halves of the double): 0xB1 = lanes [1,0,3,2], 0x4E = [2,3,0,1].
template <int CTRL>
__device__ __forceinline__ double hy_dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
)HIP";
    // (Registers: what lets `per_cu` workgroups of `wps` wavefronts share the four SIMDs of a CU.)
    // (HEYOKA_AMD_TABLE_LDS=7 / 8, A/B harness: the order loop unrolled - the terms of the convolutions become straight-line
    // code with constant LDS offsets -, 8: compiled for two wavefronts per SIMD, 256 registers.)
    const bool unroll_orders = opts.dev.table_lds == 7 || opts.dev.table_lds == 8;
    // NOTE: not more wavefronts than the registers of a lane allow WITHOUT spilling its table registers (offsets and constants
    // of its nodes, loaded once per kernel) - 512 / (table registers + ~56 of working set). The attribute is a demand on the
    // register allocator: a decomposition with 117 table registers compiled for the 7 wavefronts its LDS tapes allow spilled
    // 637 registers (1 560 B of scratch per lane) - and returned results which differed from run to run on the GPU
    // (profiles/r06_staged_spill_nondeterminism.log; the same source is exact under the emulator).
    const std::uint64_t n_table_regs = lt.urows.size() + 2u * lt.drows.size() + 2u * lt.prows.size();
    // (Rounds through the interpreter are non-inlined calls: room for the callee's registers as well.)
    const std::uint64_t n_work_regs = 56u + (n_generic != 0u ? 64u : 0u);
    const auto waves_by_regs = std::max<std::uint64_t>(1u, 512u / (((n_table_regs + n_work_regs) + 7u) / 8u * 8u));
    const auto waves_per_simd
        = opts.dev.table_lds == 8
              ? std::uint64_t(2)
              : std::max<std::uint64_t>(1u, std::min<std::uint64_t>({std::uint64_t(8), (per_cu * wps + 3u) / 4u, waves_by_regs}));
    src << "extern \"C\" __global__ __attribute__((amdgpu_waves_per_eu(" << waves_per_simd << "))) void __launch_bounds__(" << LANES
        << ") hy_taylor(const hy_kargs a)\n{\n";
    src << "const unsigned lane = threadIdx.x;\nconst u64 N = a.N;\n";
    for (std::size_t t = 0; t < lt.urows.size(); ++t) {
        src << "const unsigned ru" << t << " = hy_st_u[" << t * LANES << "u + lane];\n";
    }
    for (std::size_t t = 0; t < lt.drows.size(); ++t) {
        src << "const double rd" << t << " = hy_st_d[" << t * LANES << "u + lane];\n";
    }
    src << R"HIP(
hy_tctx c;
c.tape = nullptr;
c.T = 0;
c.pars = a.pars;
c.N = N;
#if defined(HY_STATIC_Q)
for (u64 s = blockIdx.x; s < N; s += gridDim.x) {
#else
for (;;) {
    // The next system from the device-side work queue (one atomic per workgroup).
    u64 s = 0;
#if HY_WPS > 1u
    __syncthreads();
    if (lane == 0u) hy_sh_base = atomicAdd((u64 *)(a.counters + 2), 1ull);
    __syncthreads();
    s = hy_sh_base;
#else
    if (lane == 0u) s = atomicAdd((u64 *)(a.counters + 2), 1ull);
    // (Through readfirstlane the position is a scalar for the compiler and the exit of the work loop a wave-uniform branch.)
    s = ((u64)__builtin_amdgcn_readfirstlane((unsigned)(s >> 32)) << 32) | (u64)__builtin_amdgcn_readfirstlane((unsigned)s);
#endif
    if (s >= N) break;
#endif
    c.s = s;
)HIP";
    for (std::size_t t = 0; t < lt.prows.size(); ++t) {
        src << "    const double rp" << t << " = a.pars[(u64)hy_st_p[" << t * LANES << "u + lane] * N + s];\n";
    }
    src << R"HIP(
    double t_hi = a.time_hi[s], t_lo = a.time_lo[s];
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
        t_dir = (rem.hi > 0.0) | ((rem.hi == 0.0) & (rem.lo >= 0.0));
        if (a.lim != nullptr) mdt = a.lim[s];
    } else {
        step_lim = a.lim[s];
    }
    HY_SYNC();
    for (unsigned i = lane; i < HY_N_EQ; i += HY_LANES) hy_sv(i, 0u) = a.state[(u64)i * N + s];
    HY_SYNC();
    u64 n_steps = 0, iter = 0;
    double min_h = __builtin_inf(), max_h = 0.0, last_h = 0.0;
    i64 outcome = HY_OC_SUCCESS;
    for (;;) {
        double lim;
        if (a.mode == 1) {
            hy_df m; m.lo = 0.0;
            m.hi = t_dir ? mdt : -mdt;
            const bool lt_fwd = hy_df_lt(rem, m), lt_bwd = hy_df_lt(m, rem);
            const bool rem_first = (t_dir & lt_fwd) | (!t_dir & lt_bwd);
            lim = rem_first ? rem.hi : m.hi;
        } else {
            lim = step_lim;
        }
        c.t_hi = t_hi;
        // ---- order 0 ----
)HIP";
    src << o0.str();
    src << "#if HY_N_EV > 0\nconst unsigned k_end = HY_ORDER + 1u;\n#else\nconst unsigned k_end = HY_ORDER;\n#endif\n";
    src << (unroll_orders ? "#pragma unroll\n" : "#pragma nounroll\n") << "for (unsigned k = 1; k <= HY_ORDER; ++k) {\n";
    src << svk.str();
    src << "if (k == k_end) break;\n";
    src << ok.str();
    src << "}\n";
    src << R"HIP(
        // Step size (taylor_determine_h(), src/taylor_00.cpp:102-273): infinity norms over the state variables.
        double m0 = 0.0, mo = 0.0, mom1 = 0.0;
        for (unsigned i = lane; i < HY_N_EQ; i += HY_LANES) {
            m0 = hy_max(m0, fabs(hy_lds_tape[i * HY_P]));
            mo = hy_max(mo, fabs(hy_lds_tape[i * HY_P + HY_ORDER]));
            mom1 = hy_max(mom1, fabs(hy_lds_tape[i * HY_P + HY_ORDER - 1u]));
        }
#if HY_N_EV > 0
        // (The event equations take part in the norms, src/taylor_00.cpp:209-219.)
        for (unsigned e = lane; e < HY_N_EV; e += HY_LANES) {
            m0 = hy_max(m0, fabs(hy_tp(c, 0, hy_ev_u[e])));
            mo = hy_max(mo, fabs(hy_tp(c, HY_ORDER, hy_ev_u[e])));
            mom1 = hy_max(mom1, fabs(hy_tp(c, HY_ORDER - 1u, hy_ev_u[e])));
        }
#endif
        m0 = hy_wave_max(m0);
        mo = hy_wave_max(mo);
        mom1 = hy_wave_max(mom1);
#if HY_WPS > 1u
        if ((lane & 63u) == 0u) {
            hy_red[3u * (lane >> 6)] = m0; hy_red[3u * (lane >> 6) + 1u] = mo; hy_red[3u * (lane >> 6) + 2u] = mom1;
        }
        __syncthreads();
        m0 = hy_red[0]; mo = hy_red[1]; mom1 = hy_red[2];
        for (unsigned w = 1; w < HY_WPS; ++w) {
            m0 = hy_max(m0, hy_red[3u * w]); mo = hy_max(mo, hy_red[3u * w + 1u]); mom1 = hy_max(mom1, hy_red[3u * w + 2u]);
        }
#endif
        const double num_rho = (m0 <= 1.0) ? 1.0 : m0;
        const double rho_o = hy_root(num_rho / mo, 1.0 / (double)HY_ORDER);
        const double rho_om1 = hy_root(num_rho / mom1, 1.0 / (double)(HY_ORDER - 1u));
        const double rho_m = hy_min(rho_o, rho_om1);
        double h = rho_m * HY_RHOFAC;
        h = hy_min(h, fabs(lim));
        h = (lim < 0.0) ? -h : h;

        if (a.tc != nullptr) {
            for (unsigned q = lane; q < HY_N_EQ * (HY_ORDER + 1u); q += HY_LANES) {
                const unsigned i = q / (HY_ORDER + 1u), k = q % (HY_ORDER + 1u);
                a.tc[((u64)i * (HY_ORDER + 1u) + k) * N + s] = hy_lds_tape[i * HY_P + k];
            }
        }

        if (a.mode == 4) {
            // Stepper with events (taylor_add_adaptive_step_with_events(), src/taylor_00.cpp:592-710): jets of the event
            // equations, max |x_i| and the step size; the state is updated later by the dense-output kernel.
#if HY_N_EV > 0
            for (unsigned q = lane; q < HY_N_EV * (HY_ORDER + 1u); q += HY_LANES) {
                const unsigned e = q / (HY_ORDER + 1u), k = q % (HY_ORDER + 1u);
                a.ev_tc[((u64)e * (HY_ORDER + 1u) + k) * N + s] = hy_tp(c, k, hy_ev_u[e]);
            }
#endif
            a.max_abs_state[s] = m0;
            last_h = h;
            break;
        }

        // (The stores above read the order-0 row which the update below overwrites: other lanes' variables.)
        HY_SYNC();
        bool nf = false;
        for (unsigned i = lane; i < HY_N_EQ; i += HY_LANES) {
            const double *cf = hy_lds_tape + i * HY_P;
            double res;
#if HY_HIGH_ACCURACY
            res = cf[0];
            double comp = 0.0, cur_h = h;
            for (unsigned k = 1; k <= HY_ORDER; ++k) {
                const double tmp = cf[k] * cur_h;
                const double y = tmp - comp;
                const double t = res + y;
                comp = (t - res) - y;
                res = t;
                cur_h = cur_h * h;
            }
#else
            res = cf[HY_ORDER];
            for (unsigned k = 1; k <= HY_ORDER; ++k) res = cf[HY_ORDER - k] + res * h;
#endif
            hy_sv(i, 0u) = res;
            nf = nf | !hy_finite(res);
        }
#if HY_WPS > 1u
        nf = __syncthreads_or(nf ? 1 : 0) != 0;
#else
        nf = __builtin_amdgcn_ballot_w64(nf) != 0ull;
        HY_WSYNC();
#endif
        {
            hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
            hy_df hh; hh.hi = h; hh.lo = 0.0;
            const hy_df nt = hy_df_add(tcur, hh);
            t_hi = nt.hi; t_lo = nt.lo;
        }
        last_h = h;
        nf = nf | !(hy_finite(t_hi) & hy_finite(t_lo));
        HY_STEP_TAIL(nf, lane == 0u)
    }
    // NOTE: the per-system results are stored by EVERY lane (identical values to identical addresses) instead of by
    // lane 0 alone: with a second `lane == 0` region at the bottom of the work loop next to the one of the queue at its top
    // the code object of this toolchain hung on the device (the kernel never returned; bisected with early exits).
    if (a.mode == 4) {
        a.last_h[s] = last_h;
        continue;
    }
    for (unsigned i = lane; i < HY_N_EQ; i += HY_LANES) a.state[(u64)i * N + s] = hy_lds_tape[i * HY_P];
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
)HIP";

    auto text = src.str();
    // HY_SYNC: wave-level or workgroup-level.
    const std::string sync_def = wps == 1u ? "#define HY_SYNC() HY_WSYNC()\n" : "#define HY_SYNC() __syncthreads()\n";
    const auto pos = text.find("#define HY_STAGED 1\n");
    text.insert(pos, sync_def);

    ret.source = std::move(text);
    ret.kernel_name = "hy_taylor";
    ret.dout_name = "hy_dout";
    ret.mode = emit_mode::table;
    ret.persistent = true;
    ret.tc_optional = true;
    ret.block_size = LANES;
    ret.lanes_per_system = LANES;
    ret.scratch_per_wave = 0;
    ret.n_statements = n_rounds_total;
    ret.notes = "table mode (staged): " + std::to_string(n_nodes) + " nodes in " + std::to_string(groups.size())
                + " groups / " + std::to_string(n_levels) + " dependency levels (" + std::to_string(n_rounds_total)
                + " rounds, " + std::to_string(n_generic) + " through the interpreter), one system per workgroup of "
                + std::to_string(LANES) + " lanes, tape in LDS (" + std::to_string(tape_bytes) + " B, "
                + std::to_string(std::min<std::uint64_t>(per_cu, 32u / wps)) + " systems per CU), "
                + std::to_string(lt.urows.size() + 2u * lt.drows.size() + 2u * lt.prows.size()) + " table registers per lane, " + std::to_string(n_hist) + " rows + " + std::to_string(n_slab) + " cells, " + std::to_string(std::count_if(parent.begin(), parent.end(), [](int x) { return x >= 0; })) + " followers and " + std::to_string(n_eq - n_rest) + " state-variable recursions fused, " + std::to_string(std::count(virt.begin(), virt.end(), char(1))) + " differences never stored, " + (strict ? std::string("strict order of the additions") : std::to_string(n_split) + " groups with split convolutions");
    return ret;
}

} // namespace heyoka_amd
