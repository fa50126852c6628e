// Planning structures of the cluster code generators (see hip_emit_cluster.cpp).
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "hip_emit_detail.hpp"

namespace heyoka_amd::cluster_detail
{

struct glue_group {
    std::uint32_t level = 0;
    std::string key;
    std::vector<std::uint32_t> nodes; // u indices
};

// A class of isomorphic clusters at one dependency level (multi-class plans: systems whose nonlinear sub-DAGs come in
// several shapes - point-mass pairs next to oblateness or drag terms, ... - run every class as its own section of
// straight-line code, cluster i of a class on lane i of the system's lane group).
struct cluster_class {
    std::vector<std::uint32_t> members; // cluster ids (indices into cluster_plan::clusters), the first one is the template
    std::uint32_t level = 1;
    std::vector<std::pair<std::uint32_t, std::uint32_t>> cst_pos; // (template position, arg index) of per-lane constants
    std::vector<std::vector<double>> cst_val;                     // [member][slot]
    std::vector<std::uint32_t> out_pos;                           // template positions whose value is exported
    std::vector<std::uint32_t> stored_pos;                        // template positions with a history
};

struct cluster_plan {
    std::uint32_t n_eq = 0, n_u = 0, L = 1, spw = 64;
    std::vector<std::vector<std::uint32_t>> clusters; // member u indices (ascending), all isomorphic
    std::vector<int> cluster_of;                      // per u: cluster id or -1
    std::uint32_t cluster_level = 1;
    // Template-relative descriptions.
    std::vector<std::vector<std::uint32_t>> ext_u;   // [cluster][e] -> u index of the e-th external input
    std::vector<std::pair<std::uint32_t, std::uint32_t>> cst_pos; // (template position, arg index) of per-lane constants
    std::vector<std::vector<double>> cst_val;         // [cluster][slot]
    // Parameter operands of the members whose index differs between the clusters: (template position, arg index)
    // and, per cluster, the parameter indices (loaded per lane).
    std::vector<std::pair<std::uint32_t, std::uint32_t>> par_pos;
    std::vector<std::vector<std::uint32_t>> par_idx;
    // A glue node has a parameter operand (glue groups are keyed by shape, the parameter index is per lane).
    bool glue_has_par = false;
    std::vector<std::uint32_t> out_pos;               // template positions whose value is exported
    std::vector<int> slot_of;                         // per u: LDS slot or -1
    std::uint32_t n_slots = 0, n_dummy = 0;
    std::vector<glue_group> groups;                   // sorted by level
    std::uint32_t max_level = 0;
    std::vector<std::uint32_t> lvl;                   // per u
    // Template positions of the members whose lower-order coefficients are read by a recurrence
    // (they must be kept for the whole step: in registers, or on a tape).
    std::vector<std::uint32_t> stored_pos;
    // Multi-class plans (plan_limits::multi_class): the classes; the single-class fields above then describe class 0.
    std::vector<cluster_class> classes;
};

// Limits of the target kernel shape: wave mode (one system per group of <= 64 lanes, jets in registers) or
// block mode (one system per workgroup, any number of clusters, jets on a tape).
struct plan_limits {
    std::uint32_t max_clusters = 64;
    bool jets_in_registers = true;
    // Absorb the linear nodes fed by a single cluster into that cluster (fewer glue rounds). Switched off by
    // make_plan() on a second attempt when the absorption makes otherwise isomorphic clusters differ.
    bool absorb_linear = true;
    // Parameter operands as per-lane values (the index may differ between isomorphic clusters / the nodes of a glue
    // group: cluster_plan::par_pos / par_idx, glue_has_par). Off: the index is part of the shape.
    bool generic_pars = false;
    // Several classes of clusters: isomorphic within a class, any shapes and dependency levels between the classes.
    bool multi_class = false;
};

// Build the plan; returns an empty string on success, otherwise the reason why cluster mode is not applicable.
std::string make_plan(const taylor_program &p, std::uint32_t order, cluster_plan &pl, const plan_limits &lim = {});

inline bool is_var(const operand &o)
{
    return o.type == operand::kind::uvar;
}

// Point-mass pair clusters: {d_0, d_1, d_2 = coordinate differences (sub of two external inputs), sum_sq(d_0, d_1, d_2),
// pow(sum_sq, alpha), [c * pow], d_i * pow, [c_i * (d_i * pow)]} - the shape the lane-pair (v3) and wave-role (v4)
// generators split over two lanes / two wavefronts. Template-relative positions of the members.
struct pair_pattern {
    bool ok = false;
    std::uint32_t d[3] = {}, sq = 0, pw = 0, pr[3] = {};
    int sc = -1, rx[3] = {-1, -1, -1};
    double ex = 0;
    // External inputs of the three differences: template ext indices (minuend, subtrahend).
    std::uint32_t de[3][2] = {};
};

// Match the template cluster of a plan against the pattern (pp.ok = structural match; callers add their own limits).
inline void detect_pair_pattern(const taylor_program &p, const cluster_plan &pl, pair_pattern &pp)
{
    const auto n_eq = p.n_eq;
    const auto &t0 = pl.clusters[0];

        const auto npos = static_cast<std::uint32_t>(t0.size());
        std::map<std::uint32_t, std::uint32_t> pos_of, ext_of;
        for (std::uint32_t q = 0; q < npos; ++q) {
            pos_of[t0[q]] = q;
        }
        // Template ext numbering exactly as in make_plan() (first encounter, members in order, arguments in order).
        for (std::uint32_t q = 0; q < npos; ++q) {
            for (const auto &o : p.nodes[t0[q] - n_eq].args) {
                if (is_var(o) && pos_of.count(o.idx) == 0u && ext_of.count(o.idx) == 0u) {
                    const auto e = static_cast<std::uint32_t>(ext_of.size());
                    ext_of[o.idx] = e;
                }
            }
        }
        const auto node_at = [&](std::uint32_t q) -> const dc_node & { return p.nodes[t0[q] - n_eq]; };
        const auto member = [&](const operand &o) { return is_var(o) && pos_of.count(o.idx) != 0u; };
        int q_pw = -1, n_pw = 0;
        for (std::uint32_t q = 0; q < npos; ++q) {
            const auto &n = node_at(q);
            if (n.kind == func_kind::pow && member(n.args[0]) && n.args[1].type == operand::kind::num
                && n.args[1].value != .5 && n.args[1].value != 2.) {
                q_pw = static_cast<int>(q);
                ++n_pw;
            }
        }
        bool ok = n_pw == 1;
        std::vector<char> used(npos, 0);
        if (ok) {
            pp.pw = static_cast<std::uint32_t>(q_pw);
            pp.ex = node_at(pp.pw).args[1].value;
            used[pp.pw] = 1;
            pp.sq = pos_of.at(node_at(pp.pw).args[0].idx);
            const auto &ns = node_at(pp.sq);
            ok = ns.kind == func_kind::sum_sq && ns.args.size() == 3u && used[pp.sq] == 0;
            used[pp.sq] = 1;
            for (std::uint32_t c = 0; ok && c < 3u; ++c) {
                ok = member(ns.args[c]);
                if (!ok) {
                    break;
                }
                pp.d[c] = pos_of.at(ns.args[c].idx);
                const auto &nd = node_at(pp.d[c]);
                ok = used[pp.d[c]] == 0 && nd.kind == func_kind::sub && nd.args.size() == 2u && is_var(nd.args[0])
                     && is_var(nd.args[1]) && !member(nd.args[0]) && !member(nd.args[1]);
                if (ok) {
                    used[pp.d[c]] = 1;
                    pp.de[c][0] = ext_of.at(nd.args[0].idx);
                    pp.de[c][1] = ext_of.at(nd.args[1].idx);
                }
            }
        }
        // Scaling of the pow (G * m_j * r^-3).
        for (std::uint32_t q = 0; ok && q < npos; ++q) {
            const auto &n = node_at(q);
            if (used[q] == 0 && n.kind == func_kind::prod && n.args.size() == 2u && n.args[0].type == operand::kind::num
                && !(n.args[0].value == -1.) && member(n.args[1]) && pos_of.at(n.args[1].idx) == pp.pw) {
                if (pp.sc != -1) {
                    ok = false;
                }
                pp.sc = static_cast<int>(q);
                used[q] = 1;
            }
        }
        const auto q_r = pp.sc >= 0 ? static_cast<std::uint32_t>(pp.sc) : pp.pw;
        bool have_pr[3] = {false, false, false};
        for (std::uint32_t q = 0; ok && q < npos; ++q) {
            const auto &n = node_at(q);
            if (used[q] != 0 || n.kind != func_kind::prod || n.args.size() != 2u || !member(n.args[0]) || !member(n.args[1])) {
                continue;
            }
            const auto a0 = pos_of.at(n.args[0].idx), a1 = pos_of.at(n.args[1].idx);
            const auto dq = (a0 == q_r) ? a1 : (a1 == q_r ? a0 : npos);
            for (std::uint32_t c = 0; c < 3u; ++c) {
                if (dq == pp.d[c] && !have_pr[c]) {
                    pp.pr[c] = q;
                    have_pr[c] = true;
                    used[q] = 1;
                }
            }
        }
        ok = ok && have_pr[0] && have_pr[1] && have_pr[2];
        for (std::uint32_t q = 0; ok && q < npos; ++q) {
            const auto &n = node_at(q);
            if (used[q] != 0) {
                continue;
            }
            bool hit = false;
            // (A reaction may well be a negation - equal masses -: its factor is a per-lane table value like any other.)
            if (n.kind == func_kind::prod && n.args.size() == 2u && n.args[0].type == operand::kind::num && member(n.args[1])) {
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    if (pos_of.at(n.args[1].idx) == pp.pr[c] && pp.rx[c] == -1) {
                        pp.rx[c] = static_cast<int>(q);
                        used[q] = 1;
                        hit = true;
                    }
                }
            }
            ok = hit;
        }
        // Only the products (and their scalings) may be read from outside.
        for (const auto q : pl.out_pos) {
            bool fine = false;
            for (std::uint32_t c = 0; c < 3u; ++c) {
                fine = fine || q == pp.pr[c] || (pp.rx[c] >= 0 && q == static_cast<std::uint32_t>(pp.rx[c]));
            }
            ok = ok && fine;
        }
        ok = ok && ((pp.rx[0] >= 0) == (pp.rx[1] >= 0)) && ((pp.rx[0] >= 0) == (pp.rx[2] >= 0));
    pp.ok = ok;
}

} // namespace heyoka_amd::cluster_detail
