// Cluster code generation, version 2: software-pipelined orders.
//
// Same partition as hip_emit_cluster.cpp (clusters in registers, glue through an LDS slab), with the
// per-order critical path shortened:
//   * state-variable recursions are *fused* into the glue round that produces their right-hand side
//     (x^[k+1] = rhs^[k] / (k + 1) is computed by the lane that just computed rhs^[k]; chains such as
//     x' = v, v' = a ride along), so that an order needs one LDS synchronisation per dependency level
//     (2 for N-body systems) instead of two extra ones for a separate state-variable phase;
//   * the LDS slab is double-buffered by order parity (no write-after-read hazards between orders);
//   * the history part of every convolution of order k + 1 (terms without order-(k+1) operands) is
//     emitted at the end of order k, where it overlaps the latency of the glue exchange; only the
//     1-2 terms involving new coefficients remain on the critical path (ssa_emitter::node_partial /
//     node_finish). Sums are FMA chains;
//   * divisions by the (constant) order use the exact FMA-based sequence ssa_emitter::div_const().
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <set>
#include <tuple>

#include "hip_emit_cluster_plan.hpp"
#include "hip_emit_detail.hpp"
#include "logging.hpp"

namespace heyoka_amd
{

namespace
{

// (min_lanes: the smallest number of lanes per system of the one-lane-per-pair kernel - see emit_cluster_v2().)
emitted_module emit_cluster_v2_impl(const taylor_program &p, const emit_options &opts, std::string &why_not, bool allow_one_lane,
                                    bool &one_lane_jets_in_lds, std::uint32_t min_lanes = 4)
{
    using cluster_detail::cluster_plan;
    using cluster_detail::is_var;
    using emit_detail::prelude;
    using emit_detail::rhofac;
    using emit_detail::ssa_emitter;

    emitted_module ret;
    // Experiment switches of this generator: ONE environment variable, HEYOKA_AMD_V5_OPTS, a comma-separated list of flags
    // (profiles/experiments/ab.py compares variants inside one process). Every flag switches OFF one of the round-5 items:
    //   nomsq     three accumulators for the half sums of squares (one per coordinate) instead of one;
    //   nopack2   the final evaluation of a partially filled owner slot as a full two-series pass;
    const auto v5_flag = [&opts](const char *name) {
        if (opts.dev.v5_opts.empty()) {
            return false;
        }
        std::string s = std::string(",") + opts.dev.v5_opts + ",";
        std::replace(s.begin(), s.end(), '+', ','); // ('+' separates flags where ',' separates variables: ab.py)
        return s.find(std::string(",") + name + ",") != std::string::npos;
    };
    cluster_plan pl;
    // Parameter operands are per-lane values here (e.g. kw::masses = par[...]: the pair clusters differ only by the
    // indices of the parameters they read).
    cluster_detail::plan_limits plim;
    plim.generic_pars = true;
    why_not = cluster_detail::make_plan(p, opts.order, pl, plim);
    if (!why_not.empty()) {
        return ret;
    }
    const auto cu = constant_uvars(p);

    const auto n_eq = p.n_eq, order = opts.order;
    const auto nc = static_cast<std::uint32_t>(pl.clusters.size());
    const auto &t0 = pl.clusters[0];

    // ---- 0. Lane-pair variant ("v3"): point-mass pair clusters {d_0, d_1, d_2 = coordinate differences,
    // sum_sq(d_0, d_1, d_2), pow(sum_sq, alpha), [c * pow], d_i * pow, [c_i * (d_i * pow)]} are split over TWO lanes:
    // lane A owns d_0, d_1 (their products and squares), lane B owns d_2, the sum of squares and the pow recurrence.
    // Each lane keeps 4 coefficient histories instead of 5 + and the kernel fits in 256 registers, i.e. two wavefronts
    // per SIMD: with one wavefront per SIMD every instruction of the stream - LDS, scalar, register copies - costs an
    // issue slot of the FP64 pipe (profiles/ubench/issue_rate.hip), with two they overlap with the other wavefront's
    // arithmetic. The two lanes run ONE instruction stream: the chains are matched so that the same FMA is useful
    // work on both lanes with different register contents (see emit_pair_order below).
    cluster_detail::pair_pattern pp;
    cluster_detail::detect_pair_pattern(p, pl, pp);
    const bool pp_shape_ok = pp.ok;
    {
        const bool ok = pp.ok;
        // NOTE: up to 32 pairs. With 17 .. 32 pairs there is one system per wavefront (64 lanes per system); in round 2
        // that variant did not terminate on the hardware: a finished system kept taking steps whose length was only
        // clamped to zero (a nan from the selector of a non-finite state survives the clamp), fixed by forcing h = 0.
        pp.ok = ok && 2u * nc <= 64u && p.n_par == 0u && (opts.cluster_kernel == 0 || opts.cluster_kernel >= 3);
    }
    const bool m4 = opts.event_stepper;
    // ---- 0b. One lane per pair, two wavefronts per SIMD ("v5"): the lane-pair split halves the histories a lane keeps
    // (4 x 20 doubles) so that the kernel fits in 256 registers, but every piece of work outside the convolution
    // chains - the role union of the finishing operations, the glue round, the serial tail of the step - is then paid
    // per TWO systems of a wavefront. Five histories (d_0, d_1, d_2, b / b_0, sa) of 19 entries fit in 256 registers as
    // well once the pow recurrence stops keeping j * sa_j: with T_j = sum_{i >= j} p_i (p_i = b_{k-i} sa_i) the weighted
    // sum is a sum of suffix sums, S2 = sum_j j p_j = sum_j T_j - one FMA and one addition per term, like the two FMAs
    // of the weighted form, and no sixth history. One lane per pair: 16 lanes per system for the 15 pairs of the outer
    // Solar System, FOUR systems per wavefront, no role union and no lane exchanges inside a pair. The jets of the
    // state variables of 32 systems per CU do not fit in LDS (199 KB): only the velocity-type variables (the ones
    // defined by a glue node) are stored, the coefficients of the position-type ones (x' = v) are re-derived in the final
    // evaluation as x^[k] = v^[k-1] * RN(1 / k) - the very operation which produced them.
    const bool one_lane = [&]() {
        if (!allow_one_lane) {
            return false;
        }
        // (The derived position jets rely on the reciprocal form of the division by the order.)
        // (Stepper with events: this kernel keeps exactly the COMPACT set of Taylor coefficients - velocity-type jets and
        // the current values of the position-type variables -, see emitted_module::compact_tc; HEYOKA_AMD_V5_EVENTS=0 and
        // HEYOKA_AMD_COMPACT_TC=0 keep the stepper with events on the lane-pair kernel.)
        const bool m4_ok = !m4 || (opts.dev.v5_events && opts.dev.compact_tc);
        return pp_shape_ok && p.n_par == 0u && m4_ok && nc <= 64u && !opts.exact_division;
    }();
    if (one_lane) {
        pp.ok = false;
    }
    const bool pair_split = pp.ok;
    // (Shared by the two pair-pattern kernels: fused reactions, merged schedule, reciprocal-based divisions.)
    const bool pairk = pair_split || one_lane;
    if (pair_split) {
        pl.L = 2;
        while (pl.L < 2u * nc) {
            pl.L *= 2u;
        }
        pl.spw = 64u / pl.L;
        // Slab slots of the cluster outputs in lane order: the lanes of a group write consecutive doubles (distinct LDS
        // banks). d_0 * sa / d_2 * sa alternate (lane = 2 * pair + role), d_1 * sa is written by the A lanes only.
        std::uint32_t ns = n_eq;
        std::fill(pl.slot_of.begin() + n_eq, pl.slot_of.end(), -1);
        const auto number = [&](const auto &member_of_lane, std::uint32_t n_lanes) {
            for (std::uint32_t l = 0; l < n_lanes; ++l) {
                pl.slot_of[member_of_lane(l)] = static_cast<int>(ns++);
            }
        };
        number([&](std::uint32_t l) { return pl.clusters[l / 2u][pp.pr[(l & 1u) != 0u ? 2u : 0u]]; }, 2u * nc);
        number([&](std::uint32_t c) { return pl.clusters[c][pp.pr[1]]; }, nc);
        if (pp.rx[0] >= 0) {
            number([&](std::uint32_t l) {
                return pl.clusters[l / 2u][static_cast<std::uint32_t>(pp.rx[(l & 1u) != 0u ? 2u : 0u])];
            }, 2u * nc);
            number([&](std::uint32_t c) { return pl.clusters[c][static_cast<std::uint32_t>(pp.rx[1])]; }, nc);
        }
        for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
            if (pl.cluster_of[u] == -1) {
                pl.slot_of[u] = static_cast<int>(ns++);
            }
        }
        pl.n_slots = ns;
    }
    if (one_lane) {
        pl.L = min_lanes;
        while (pl.L < nc) {
            pl.L *= 2u;
        }
        pl.spw = 64u / pl.L;
        // Outputs in lane order, one run of consecutive slots per product.
        std::uint32_t ns = n_eq;
        std::fill(pl.slot_of.begin() + n_eq, pl.slot_of.end(), -1);
        // (Numbered again after the compaction below: [lane][product] with a stride of 3 doubles between the lanes of a
        // group - the 16 lanes which a ds_write_b64 services together hit 16 different bank pairs -, the reactions in a
        // second region of the same shape.)
        for (std::uint32_t c = 0; c < nc; ++c) {
            for (std::uint32_t i = 0; i < 3u; ++i) {
                pl.slot_of[pl.clusters[c][pp.pr[i]]] = static_cast<int>(ns++);
            }
        }
        for (std::uint32_t c = 0; pp.rx[0] >= 0 && c < nc; ++c) {
            for (std::uint32_t i = 0; i < 3u; ++i) {
                pl.slot_of[pl.clusters[c][static_cast<std::uint32_t>(pp.rx[i])]] = static_cast<int>(ns++);
            }
        }
        for (std::uint32_t u = n_eq; u < p.n_u; ++u) {
            if (pl.cluster_of[u] == -1) {
                pl.slot_of[u] = static_cast<int>(ns++);
            }
        }
        pl.n_slots = ns;
    }
    const auto L = pl.L, spw = pl.spw;
    // (Lane pairs / one lane per pair: 512 threads = two wavefronts per SIMD.)
    const std::uint32_t bs = pairk ? 512u : 256u;
    const std::uint32_t wpb = bs / 64u;
    const auto n_ext = static_cast<std::uint32_t>(pl.ext_u[0].size());
    const auto n_out = static_cast<std::uint32_t>(pl.out_pos.size());
    const auto n_cst = static_cast<std::uint32_t>(pl.cst_pos.size());

    // ---- 0c. One lane per pair: reactions fused into the acceleration sums ("frx", round 5). The sensitivity experiment
    // (profiles/r05_sensitivity_marginal_costs.log) prices an LDS store at ~30 cycles of the issuing wavefront's time, five
    // FMAs; 3 of the 8 stores of an order are the reactions c * (d_i * sa), values which differ from the direct products
    // by a per-pair factor. Dropping them (timing experiment "norx", profiles/r05_ab_fused_reactions_estimate.log) is worth
    // +7.8 %. Here the sums of the first glue round read the direct product wherever they read a reaction and apply the
    // factor themselves, t_a = c_a * p_a with per-lane coefficients (c, or 1.0 for a direct product: exact, so that without
    // FMA contraction every term is rounded like the separate node); the additions keep the reference's pairwise order.
    // Registers are what this costs (five coefficient doubles per lane through the orders), so the coefficients exist only
    // once: the nodes of the sum group are reordered so that the sums made of direct products alone (the first body of every
    // pair it takes part in) come last - the rounds after the first are then plain sums without coefficients.
    bool frx = false;
    if (one_lane && !m4 && pp.rx[0] >= 0 && pl.groups.size() == 1u && !v5_flag("nofrx")) {
        auto &nodes = pl.groups[0].nodes;
        std::map<std::uint32_t, bool> is_rx_out; // cluster output -> is it a reaction?
        for (std::uint32_t c = 0; c < nc; ++c) {
            for (std::uint32_t i = 0; i < 3u; ++i) {
                is_rx_out[pl.clusters[c][pp.pr[i]]] = false;
                is_rx_out[pl.clusters[c][static_cast<std::uint32_t>(pp.rx[i])]] = true;
            }
        }
        const auto &n0 = p.nodes[nodes[0] - n_eq];
        frx = n0.kind == func_kind::sum && n0.args.size() >= 2u;
        std::vector<char> plain(nodes.size(), 1);
        for (std::size_t j = 0; j < nodes.size() && frx; ++j) {
            const auto &nd = p.nodes[nodes[j] - n_eq];
            frx = nd.kind == func_kind::sum && nd.args.size() == n0.args.size();
            for (const auto &o : nd.args) {
                const auto it = is_var(o) ? is_rx_out.find(o.idx) : is_rx_out.end();
                frx = frx && it != is_rx_out.end();
                if (frx && it->second) {
                    plain[j] = 0;
                }
            }
        }
        if (frx) {
            // (Stable: the three coordinates of a body stay adjacent and in order, which the velocity exchange relies on.)
            std::vector<std::uint32_t> first, last;
            for (std::size_t j = 0; j < nodes.size(); ++j) {
                (plain[j] != 0 ? last : first).push_back(nodes[j]);
            }
            // Every round after the first must consist of plain sums.
            frx = first.size() <= pl.L && !last.empty();
            if (frx) {
                first.insert(first.end(), last.begin(), last.end());
                nodes = std::move(first);
            }
        }
    }

    // ---- 1. Anchor every state variable to the glue node at the root of its rhs chain. ----
    // anchor[i] = glue u variable, depth[i] >= 1.
    std::vector<int> anchor(n_eq, -1);
    std::vector<std::uint32_t> depth(n_eq, 0);
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        std::uint32_t cur = i, d = 0;
        std::set<std::uint32_t> seen;
        for (;;) {
            if (!seen.insert(cur).second) {
                why_not = "cyclic chain of state-variable definitions";
                return ret;
            }
            const auto &def = p.sv_defs[cur];
            ++d;
            if (def.type != operand::kind::uvar) {
                why_not = "a state variable is defined by a constant or a parameter";
                return ret;
            }
            if (def.idx < n_eq) {
                cur = def.idx;
                continue;
            }
            if (pl.cluster_of[def.idx] != -1) {
                why_not = "a state variable is defined directly by a cluster member";
                return ret;
            }
            anchor[i] = static_cast<int>(def.idx);
            depth[i] = d;
            break;
        }
    }
    // att[u] = state variables anchored at glue node u, sorted by depth.
    std::map<std::uint32_t, std::vector<std::uint32_t>> att;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        att[static_cast<std::uint32_t>(anchor[i])].push_back(i);
    }
    for (auto &[u, v] : att) {
        std::sort(v.begin(), v.end(), [&](std::uint32_t a, std::uint32_t b) { return depth[a] < depth[b]; });
        for (std::size_t j = 0; j < v.size(); ++j) {
            // Chain shape: depths 1, 2, 3, ... each defined by the previous one (or by the anchor).
            if (depth[v[j]] != j + 1u) {
                why_not = "branching state-variable chains";
                return ret;
            }
            const auto &def = p.sv_defs[v[j]];
            const auto expect = (j == 0u) ? u : v[j - 1u];
            if (def.idx != expect) {
                why_not = "branching state-variable chains";
                return ret;
            }
        }
    }
    // All the nodes of a glue group must carry the same number of attached variables.
    std::vector<std::uint32_t> grp_natt(pl.groups.size(), 0);
    for (std::size_t g = 0; g < pl.groups.size(); ++g) {
        const auto &nodes = pl.groups[g].nodes;
        const auto it0 = att.find(nodes[0]);
        const auto n0 = it0 == att.end() ? 0u : static_cast<std::uint32_t>(it0->second.size());
        for (const auto u : nodes) {
            const auto it = att.find(u);
            const auto n = it == att.end() ? 0u : static_cast<std::uint32_t>(it->second.size());
            if (n != n0) {
                why_not = "glue nodes of one group with different state-variable chains";
                return ret;
            }
        }
        grp_natt[g] = n0;
    }

    // Renumber the slab slots of the state variables in owner order (group, round, chain position, lane), so
    // that the lanes publishing new coefficients write consecutive slots (distinct LDS banks).
    {
        std::uint32_t next = 0;
        for (std::size_t g = 0; g < pl.groups.size(); ++g) {
            const auto &nodes = pl.groups[g].nodes;
            const auto n_nodes = static_cast<std::uint32_t>(nodes.size());
            for (std::uint32_t r = 0; r * L < n_nodes; ++r) {
                for (std::uint32_t a = 0; a < grp_natt[g]; ++a) {
                    for (std::uint32_t l = 0; l < L && r * L + l < n_nodes; ++l) {
                        pl.slot_of[att.at(nodes[r * L + l])[a]] = static_cast<int>(next++);
                    }
                }
            }
        }
        if (next != n_eq) {
            why_not = "internal error: state-variable slots";
            return ret;
        }
    }

    // Reaction fusion (lane-pair variant): the members c * pr of a cluster (the reaction on the second body of the pair)
    // are not computed / exported by the cluster lanes: the glue sums which read them read the direct product pr instead
    // and multiply it by a per-lane coefficient (c, or 1.0 where the sum reads the direct product itself: exact, so the
    // rounding sequence of every term is the one of the separate node). Two LDS stores less per order and lane.
    // rx_fused[u] = 1 for the fused members; rx_src[u] = the product they scale.
    std::vector<char> rx_fused(p.n_u, 0);
    std::vector<std::uint32_t> rx_src(p.n_u, 0);
    bool fuse_rx = false;
    if (pairk && pp.rx[0] >= 0) {
        // (One-lane pair kernel: the reactions are computed and exported by the pair lane - the slab of a system is
        // single-buffered there and has room for them - so that the sums need no per-lane coefficients: 20 registers.)
        fuse_rx = !one_lane || frx;
        for (std::size_t c = 0; c < nc; ++c) {
            for (std::uint32_t i = 0; i < 3u; ++i) {
                const auto u = pl.clusters[c][static_cast<std::uint32_t>(pp.rx[i])];
                rx_fused[u] = 1;
                rx_src[u] = pl.clusters[c][pp.pr[i]];
            }
        }
        std::vector<char> in_sum_group(p.n_u, 0);
        for (const auto &g : pl.groups) {
            const auto &n0 = p.nodes[g.nodes[0] - n_eq];
            const bool all_var = std::all_of(n0.args.begin(), n0.args.end(), [](const operand &o) { return is_var(o); });
            if (n0.kind == func_kind::sum && all_var) {
                for (const auto u : g.nodes) {
                    in_sum_group[u] = 1;
                }
            }
        }
        for (std::uint32_t u = n_eq; fuse_rx && u < p.n_u; ++u) {
            for (const auto &o : p.nodes[u - n_eq].args) {
                if (is_var(o) && rx_fused[o.idx] != 0 && in_sum_group[u] == 0) {
                    fuse_rx = false;
                }
            }
        }
        for (const auto &d : p.sv_defs) {
            if (is_var(d) && rx_fused[d.idx] != 0) {
                fuse_rx = false;
            }
        }
        if (!fuse_rx) {
            std::fill(rx_fused.begin(), rx_fused.end(), 0);
        }
    }
    frx = frx && fuse_rx;
    // A glue node / state variable needs a slab slot only if somebody reads it through the slab.
    std::vector<char> glue_read(p.n_u, 0);
    for (const auto &n : p.nodes) {
        for (const auto &o : n.args) {
            if (is_var(o)) {
                glue_read[o.idx] = 1;
            }
        }
    }
    std::vector<std::uint32_t> lane_pr, lane_rx; // one-lane pair kernel: output slot triples of the lanes
    bool wide_rd = false;                        // ... wide-read layout: slots arranged by consumer (see below)
    // ... velocity exchange: the pair lanes take the coordinate differences from the velocity jets, d^[k] = (v_a^[k-1] - v_b^[k-1])
    // RN(1 / k), and no position coefficient is published (see below). vx_col[l] = first jet column of the two bodies of lane l.
    bool vexch = false;
    std::vector<std::array<std::uint32_t, 2>> vx_col;
    std::vector<std::array<std::uint32_t, 3>> wide_pr, wide_rx; // ... its output slots, per lane and coordinate
    std::uint32_t slab_stride_opt = 0;
    std::uint64_t bank_cost = 0;
    // One-lane pair kernel: the bookkeeping block of a system (16 doubles parked between the tails of two steps) sits at the
    // end of its slab instead of in an array of its own - its address is the slab pointer + a constant, and the register
    // which held it (spilled: two scratch reloads per step) is gone: +0.7 %, profiles/r05_ab_bookkeeping_in_slab.log (a
    // laundered system index in the retire block, against hoisted address arithmetic, measured nothing).
    const bool bk_in_slab = one_lane && !v5_flag("nobkslab");
    if (one_lane) {
        // 32 systems per CU: the slab only keeps the slots which are read through it (positions, products, glue nodes
        // with readers): 63 instead of 144 for the outer Solar System.
        for (std::size_t g = 0; g < pl.groups.size(); ++g) {
            if (grp_natt[g] > 2u) {
                why_not = "one-lane pair kernel: state-variable chains longer than two";
                return ret;
            }
        }
        // ---- Wide-read layout (round 5). A wavefront pays ~7 cycles of its own time for every LDS instruction it issues
        // (profiles/README.md, issue-rate table), and a round of the step reads 6 positions + 2 x 5 sum operands with 16
        // ds_read_b64. Slots arranged BY CONSUMER make them 10 reads: the three coordinates of a body are adjacent
        // ([x, y, z, -]: one ds_read_b128 + one ds_read_b64 per body), and the five operands of an acceleration sum are
        // adjacent ([t0 .. t4, -]: two ds_read_b128 + one ds_read_b64 per sum, pairs (t0, t1), (t2, t3) as the pairwise sum
        // takes them). A ds_read_b128 moves its 16 bytes per lane at the same LDS-array rate as two ds_read_b64
        // (MI355X_MICROARCH.md, LDS table). The products / reactions of a pair lane now go to the operand arrays of the
        // sums which read them: slot = array of (coordinate i, body) + operand index, with the arrays laid out
        // [coordinate][body] so that the three stores of a lane differ by a constant (one table register for the three).
        // The unused sixth slot of the arrays of the first two bodies takes the stores of the idle lanes. The order of the
        // distance between the slabs of two systems comes out of a scan with the bank model of the guide (reads in the lane
        // groups of each instruction width).
        wide_rd = !v5_flag("nowide") && pp.rx[0] >= 0 && pl.groups.size() == 1u;
        std::vector<std::array<std::uint32_t, 3>> bodies; // position variables (x, y, z) of every body
        std::vector<std::uint32_t> node_coord, node_rank;   // per node of the glue group
        std::uint32_t n_rank = 0, n_args = 0;
        if (wide_rd) {
            std::map<std::uint32_t, std::pair<std::uint32_t, std::uint32_t>> pos_of; // position variable -> (body, coordinate)
            for (std::uint32_t c = 0; c < nc && wide_rd; ++c) {
                for (std::uint32_t sd = 0; sd < 2u; ++sd) {
                    std::array<std::uint32_t, 3> tr{};
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        tr[i] = pl.ext_u[c][pp.de[i][sd]];
                    }
                    auto it = std::find(bodies.begin(), bodies.end(), tr);
                    if (it == bodies.end()) {
                        bodies.push_back(tr);
                        it = bodies.end() - 1;
                    }
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        const auto key = std::make_pair(static_cast<std::uint32_t>(it - bodies.begin()), i);
                        const auto ins = pos_of.emplace(tr[i], key);
                        wide_rd = wide_rd && ins.first->second == key && tr[i] < n_eq && glue_read[tr[i]] != 0;
                    }
                }
            }
            // The sums: one group, all arguments exported cluster outputs, every output read by exactly one sum.
            const auto &grp = pl.groups[0];
            const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
            n_args = static_cast<std::uint32_t>(n0.args.size());
            wide_rd = wide_rd && n0.kind == func_kind::sum && n_args >= 2u && n_args <= 6u;
            std::map<std::uint32_t, std::uint32_t> out_coord; // cluster output -> coordinate
            std::map<std::uint32_t, std::pair<std::uint32_t, bool>> out_src; // cluster output -> (cluster, is reaction)
            for (std::uint32_t c = 0; c < nc; ++c) {
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    out_coord[pl.clusters[c][pp.pr[i]]] = i;
                    out_src[pl.clusters[c][pp.pr[i]]] = {c, false};
                    out_coord[pl.clusters[c][static_cast<std::uint32_t>(pp.rx[i])]] = i;
                    out_src[pl.clusters[c][static_cast<std::uint32_t>(pp.rx[i])]] = {c, true};
                }
            }
            std::map<std::uint32_t, std::uint32_t> n_readers;
            node_coord.assign(grp.nodes.size(), 0);
            node_rank.assign(grp.nodes.size(), 0);
            // (Signature of a sum: its operands as (cluster, kind) in argument order - the sums of the three coordinates of a
            // body have the same one.)
            std::vector<std::vector<std::pair<std::uint32_t, bool>>> sig(grp.nodes.size());
            for (std::size_t j = 0; j < grp.nodes.size() && wide_rd; ++j) {
                const auto &nd = p.nodes[grp.nodes[j] - n_eq];
                wide_rd = wide_rd && nd.args.size() == n_args;
                for (std::uint32_t a = 0; a < nd.args.size() && wide_rd; ++a) {
                    const auto &o = nd.args[a];
                    wide_rd = wide_rd && is_var(o) && out_coord.count(o.idx) != 0u;
                    if (!wide_rd) {
                        break;
                    }
                    if (a == 0u) {
                        node_coord[j] = out_coord[o.idx];
                    }
                    wide_rd = wide_rd && out_coord[o.idx] == node_coord[j];
                    ++n_readers[o.idx];
                    sig[j].push_back(out_src[o.idx]);
                }
            }
            for (const auto &[u, cnt] : n_readers) {
                (void)u;
                wide_rd = wide_rd && cnt == 1u;
            }
            // Every output which has a slot must have a reader (the others are not stored).
            for (const auto &[u, src_] : out_src) {
                (void)src_;
                wide_rd = wide_rd && (pl.slot_of[u] < 0 || n_readers.count(u) != 0u);
            }
            std::vector<std::vector<std::pair<std::uint32_t, bool>>> ranks;
            for (std::size_t j = 0; j < grp.nodes.size() && wide_rd; ++j) {
                auto it = std::find(ranks.begin(), ranks.end(), sig[j]);
                if (it == ranks.end()) {
                    ranks.push_back(sig[j]);
                    it = ranks.end() - 1;
                }
                node_rank[j] = static_cast<std::uint32_t>(it - ranks.begin());
            }
            n_rank = static_cast<std::uint32_t>(ranks.size());
            // (Three sums per rank, one per coordinate; at least two ranks for the dummy stores.)
            wide_rd = wide_rd && n_rank >= 2u && grp.nodes.size() == 3u * n_rank;
            for (std::uint32_t r = 0; r < n_rank && wide_rd; ++r) {
                std::uint32_t seen = 0;
                for (std::size_t j = 0; j < grp.nodes.size(); ++j) {
                    if (node_rank[j] == r) {
                        seen |= 1u << node_coord[j];
                    }
                }
                wide_rd = wide_rd && seen == 7u;
            }
        }
        // Velocity exchange (round 5). An LDS store is the most expensive instruction of this kernel: 35-40 cycles of the
        // issuing wavefront's time against 5.6 for an FMA (profiles/r05_sensitivity_marginal_costs.log: the store path of
        // a CU moves one ds_write_b64 of a wavefront per ~6 cycles and all eight wavefronts queue on it), and a round stores
        // 3 products + 3 reactions + 2 velocity coefficients (jets) + 2 position coefficients (slab). The position
        // coefficient is the velocity coefficient times RN(1 / (k + 1)): the pair lanes read the velocity jets of their
        // two bodies instead and form d^[k] = (v_a^[k-1] - v_b^[k-1]) RN(1 / k) themselves (one multiplication per
        // coordinate; the order-0 differences come from the current positions), so that the glue lanes neither compute
        // nor store x^[k+1]: 8 stores per round instead of 10. (Rounded once after the subtraction instead of once per body
        // before it: the same quantity to rounding, more accurate where the velocities are close.) Requires the jet
        // columns of the three coordinates of a body to be adjacent: the rows are laid out [system][column] here.
        if (wide_rd && !m4 && !v5_flag("novx")) {
            // Jet column of a velocity variable = index of its glue node in the group (owner slots are filled in that order).
            const auto &grp = pl.groups[0];
            std::map<std::uint32_t, std::uint32_t> col_of_pos; // position variable -> jet column of its velocity
            for (std::size_t j = 0; j < grp.nodes.size(); ++j) {
                const auto it = att.find(grp.nodes[j]);
                if (it != att.end() && it->second.size() == 2u) {
                    col_of_pos[it->second[1]] = static_cast<std::uint32_t>(j);
                }
            }
            vexch = true;
            std::vector<std::uint32_t> body_col(bodies.size(), 0);
            for (std::size_t b = 0; b < bodies.size() && vexch; ++b) {
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    const auto it = col_of_pos.find(bodies[b][i]);
                    vexch = vexch && it != col_of_pos.end() && (i == 0u || it->second == body_col[b] + i);
                    if (vexch && i == 0u) {
                        body_col[b] = it->second;
                    }
                }
            }
            if (vexch) {
                vx_col.assign(pl.L, {});
                for (std::uint32_t l = 0; l < pl.L; ++l) {
                    const auto c = l < nc ? l : 0u;
                    for (std::uint32_t sd = 0; sd < 2u; ++sd) {
                        std::array<std::uint32_t, 3> tr{};
                        for (std::uint32_t i = 0; i < 3u; ++i) {
                            tr[i] = pl.ext_u[c][pp.de[i][sd]];
                        }
                        vx_col[l][sd] = body_col[static_cast<std::size_t>(std::find(bodies.begin(), bodies.end(), tr) - bodies.begin())];
                    }
                }
            }
        }
        // (The consumer-arranged slots exclude the fused reactions - a product then has two readers -; the analysis above still
        // serves the velocity exchange.)
        const bool wide_an = wide_rd;
        wide_rd = wide_rd && !frx;
        vexch = vexch && wide_an;
        std::vector<int> remap(pl.n_slots, -1);
        std::uint32_t ns = 0;
        const auto keep = [&](std::uint32_t u) {
            if (pl.slot_of[u] < 0) {
                return false;
            }
            if (pl.cluster_of[u] != -1) {
                // Cluster outputs: the direct products, or - when only the reaction was exported - that one.
                if (fuse_rx && rx_fused[u] != 0) {
                    return pl.slot_of[rx_src[u]] < 0;
                }
                return true;
            }
            // (Velocity exchange: nobody reads a position coefficient through the slab.)
            if (vexch && u < n_eq) {
                return false;
            }
            return glue_read[u] != 0;
        };
        // (The kept slots keep their relative order.)
        std::vector<char> kept(p.n_u, 0);
        std::vector<std::pair<int, std::uint32_t>> order_v;
        for (std::uint32_t u = 0; u < p.n_u; ++u) {
            kept[u] = keep(u) ? 1 : 0;
            if (kept[u] != 0) {
                order_v.emplace_back(pl.slot_of[u], u);
            }
        }
        std::sort(order_v.begin(), order_v.end());
        for (std::uint32_t u = 0; u < p.n_u; ++u) {
            if (kept[u] == 0) {
                pl.slot_of[u] = -1;
            }
        }
        for (const auto &[old_slot, u] : order_v) {
            (void)old_slot;
            if (pl.cluster_of[u] == -1) {
                pl.slot_of[u] = static_cast<int>(ns++);
            }
        }
        // Cluster outputs: every lane (the idle ones too) owns 3 + 3 slots, read or not: lane l keeps its products in the
        // slots out_base + 3 * lane_pr[l] + i and its reactions in rx_base + 3 * lane_rx[l] + i, where lane_pr / lane_rx are
        // permutations of the lanes. The kernel is within 25 % of the LDS throughput, so the permutations and the
        // distance between the slabs of two systems are chosen to minimise the bank conflicts of the exchange (a small
        // deterministic local search over the access patterns of a step; the model is the one of the microarchitecture
        // guide: a ds_read_b64 services the lanes 0-31 / 32-63 together, bank pair = double index mod 32, every further
        // distinct address on a bank pair costs a cycle; a ds_write_b64 services 16 consecutive lanes, double index mod 16).
        const auto out_base = ns;
        const auto rx_base = out_base + 3u * pl.L;
        lane_pr.resize(pl.L);
        lane_rx.resize(pl.L);
        for (std::uint32_t l = 0; l < pl.L; ++l) {
            lane_pr[l] = lane_rx[l] = l;
        }
        // (Fused reactions: the sums read the products; no second region.)
        const bool rx_region = pp.rx[0] >= 0 && !fuse_rx;
        ns = out_base + 3u * pl.L * (rx_region ? 2u : 1u);
        const auto assign = [&]() {
            for (std::uint32_t c = 0; c < nc; ++c) {
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    pl.slot_of[pl.clusters[c][pp.pr[i]]] = static_cast<int>(out_base + 3u * lane_pr[c] + i);
                    if (rx_region) {
                        pl.slot_of[pl.clusters[c][static_cast<std::uint32_t>(pp.rx[i])]] = static_cast<int>(rx_base + 3u * lane_rx[c] + i);
                    }
                }
            }
        };
        // Read patterns of a step: per LDS read instruction, the u variable every lane of a group reads.
        std::vector<std::vector<std::uint32_t>> rd_pat;
        for (std::uint32_t i = 0; i < 3u; ++i) {
            for (std::uint32_t sd = 0; sd < 2u; ++sd) {
                std::vector<std::uint32_t> v(pl.L);
                for (std::uint32_t l = 0; l < pl.L; ++l) {
                    v[l] = pl.ext_u[l < nc ? l : 0u][pp.de[i][sd]];
                }
                rd_pat.push_back(std::move(v));
            }
        }
        for (const auto &grp : pl.groups) {
            const auto n_nodes = static_cast<std::uint32_t>(grp.nodes.size());
            const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
            for (std::uint32_t r = 0; r * pl.L < n_nodes; ++r) {
                for (std::size_t a = 0; a < n0.args.size(); ++a) {
                    if (!is_var(n0.args[a])) {
                        continue;
                    }
                    std::vector<std::uint32_t> v(pl.L);
                    for (std::uint32_t l = 0; l < pl.L; ++l) {
                        const auto j = r * pl.L + l;
                        v[l] = p.nodes[grp.nodes[j < n_nodes ? j : r * pl.L] - n_eq].args[a].idx;
                        if (fuse_rx && rx_fused[v[l]] != 0) {
                            v[l] = rx_src[v[l]];
                        }
                    }
                    rd_pat.push_back(std::move(v));
                }
            }
        }
        const auto cost = [&](std::uint32_t stride) {
            std::uint64_t tot = 0;
            const auto spw_ = 64u / pl.L;
            // Reads: two groups of 32 lanes.
            for (const auto &v : rd_pat) {
                for (std::uint32_t g = 0; g < 2u; ++g) {
                    std::map<std::uint32_t, std::set<std::uint32_t>> banks;
                    for (std::uint32_t lane = 32u * g; lane < 32u * g + 32u; ++lane) {
                        const auto q = (lane / pl.L) % spw_, l = lane % pl.L;
                        const auto a = q * stride + static_cast<std::uint32_t>(std::max(0, pl.slot_of[v[l]]));
                        banks[a % 32u].insert(a);
                    }
                    std::size_t mx = 1;
                    for (const auto &[b, st_] : banks) {
                        mx = std::max(mx, st_.size());
                    }
                    tot += mx - 1u;
                }
            }
            // Writes of the outputs: four groups of 16 lanes, three coordinates, two kinds.
            for (std::uint32_t kind = 0; kind < (rx_region ? 2u : 1u); ++kind) {
                for (std::uint32_t g = 0; g < 4u; ++g) {
                    std::map<std::uint32_t, std::set<std::uint32_t>> banks;
                    for (std::uint32_t lane = 16u * g; lane < 16u * g + 16u; ++lane) {
                        const auto q = (lane / pl.L) % spw_, l = lane % pl.L;
                        const auto a = q * stride + (kind == 0u ? out_base + 3u * lane_pr[l] : rx_base + 3u * lane_rx[l]);
                        banks[a % 16u].insert(a);
                    }
                    std::size_t mx = 1;
                    for (const auto &[b, st_] : banks) {
                        mx = std::max(mx, st_.size());
                    }
                    tot += 3u * 2u * (mx - 1u); // (a conflicting store costs two LDS cycles more, three coordinates)
                }
            }
            return tot;
        };
        // Total slots incl. the dummy area (as computed below) and the 16 bookkeeping doubles of the system, which sit at the
        // end of its slab (one address register for both: see bk_in_slab).
        const auto n_tot_est = ns + std::max<std::uint32_t>(static_cast<std::uint32_t>(pl.out_pos.size()), 6u) + (bk_in_slab ? 16u : 0u);
        assign();
        slab_stride_opt = n_tot_est;
        auto best = cost(slab_stride_opt);
        for (std::uint32_t st_ = n_tot_est; st_ < n_tot_est + 32u; ++st_) {
            if (const auto c = cost(st_); c < best) {
                best = c;
                slab_stride_opt = st_;
            }
        }
        {
            std::uint64_t rng = 0x9E3779B97F4A7C15ull;
            const auto next = [&]() {
                rng ^= rng << 13;
                rng ^= rng >> 7;
                rng ^= rng << 17;
                return rng;
            };
            for (int it = 0; it < 4000 && best != 0u; ++it) {
                auto &perm = (rx_region && (next() & 1u) != 0u) ? lane_rx : lane_pr;
                const auto i1 = static_cast<std::uint32_t>(next() % pl.L), i2 = static_cast<std::uint32_t>(next() % pl.L);
                if (i1 == i2) {
                    continue;
                }
                std::swap(perm[i1], perm[i2]);
                assign();
                auto c = cost(slab_stride_opt);
                auto cs = slab_stride_opt;
                if (it % 16 == 0) {
                    for (std::uint32_t st_ = n_tot_est; st_ < n_tot_est + 32u; ++st_) {
                        if (const auto c2 = cost(st_); c2 < c) {
                            c = c2;
                            cs = st_;
                        }
                    }
                }
                if (c <= best) {
                    best = c;
                    slab_stride_opt = cs;
                } else {
                    std::swap(perm[i1], perm[i2]);
                }
            }
            assign();
        }
        bank_cost = best;
        pl.n_slots = ns;

        if (wide_rd) {
            const auto W = (n_args + 2u) & ~1u; // slots of an operand array (>= one spare slot, even)
            const auto nb = static_cast<std::uint32_t>(bodies.size());
            const auto &grp = pl.groups[0];
            // Address lists of the LDS instructions of a round (doubles, relative to the slab of the system), per lane of a
            // system; filled for a given layout by addr_lists().
            struct lds_op {
                int width; // 16: ds_read_b128, 8: ds_read_b64, -8: ds_write_b64
                std::vector<std::uint32_t> addr;
            };
            std::vector<std::uint32_t> perm(n_rank);
            for (std::uint32_t r = 0; r < n_rank; ++r) {
                perm[r] = r;
            }
            // (Velocity exchange: no position slots.)
            const auto pos_sz = vexch ? 0u : 4u * nb;
            std::uint32_t Dd = W * n_rank, pos_base = 0, op_base = pos_sz;
            // Position of operand a of the sums of rank r inside a coordinate block: arr_pos[r][a]. Default: arrays of W slots
            // one after the other; replaced below by a placement without store conflicts where one exists.
            std::vector<std::vector<std::uint32_t>> arr_pos(n_rank, std::vector<std::uint32_t>(n_args));
            for (std::uint32_t r = 0; r < n_rank; ++r) {
                for (std::uint32_t a = 0; a < n_args; ++a) {
                    arr_pos[r][a] = W * r + a;
                }
            }
            std::uint32_t dummy_pos[2] = {W - 1u, 2u * W - 1u}; // (spare slots of the arrays of rank 0 / rank 1)
            std::uint32_t blk_used = W * n_rank;                 // (slots of a coordinate block)
            const auto op_slot = [&](std::uint32_t coord, std::uint32_t rank, std::uint32_t a) {
                return op_base + Dd * coord + arr_pos[rank][a];
            };
            const auto out_slot = [&](std::uint32_t u) {
                // (u: an exported cluster output: the slot of the operand position which reads it.)
                for (std::size_t j = 0; j < grp.nodes.size(); ++j) {
                    const auto &nd = p.nodes[grp.nodes[j] - n_eq];
                    for (std::uint32_t a = 0; a < n_args; ++a) {
                        if (nd.args[a].idx == u) {
                            return op_slot(node_coord[j], node_rank[j], a);
                        }
                    }
                }
                return 0u;
            };
            // (Dummy slots of the idle lanes: the spare slot of the arrays of rank 0 (products) and rank 1 (reactions).)
            const auto dummy_pr = [&](std::uint32_t i) { return op_base + Dd * i + dummy_pos[0]; };
            const auto dummy_rx = [&](std::uint32_t i) { return op_base + Dd * i + dummy_pos[1]; };
            // Which operand position reads the outputs of every cluster (independent of the layout parameters).
            std::vector<std::array<std::uint32_t, 3>> pr_ref(nc), rx_ref(nc); // (node index j, argument) packed: j * 8 + a
            for (std::uint32_t c = 0; c < nc; ++c) {
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    for (int kind = 0; kind < 2; ++kind) {
                        const auto u = pl.clusters[c][kind == 0 ? pp.pr[i] : static_cast<std::uint32_t>(pp.rx[i])];
                        std::uint32_t ref = ~0u;
                        for (std::size_t j = 0; j < grp.nodes.size(); ++j) {
                            const auto &nd = p.nodes[grp.nodes[j] - n_eq];
                            for (std::uint32_t a = 0; a < n_args; ++a) {
                                if (nd.args[a].idx == u) {
                                    ref = static_cast<std::uint32_t>(j) * 8u + a;
                                }
                            }
                        }
                        (kind == 0 ? pr_ref : rx_ref)[c][i] = ref;
                    }
                }
            }
            // (The three stores of a lane must differ by the block distance: same rank and argument for the three coordinates.)
            for (std::uint32_t c = 0; c < nc && wide_rd; ++c) {
                for (const auto *ref : {&pr_ref, &rx_ref}) {
                    const auto r0 = (*ref)[c][0];
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        const auto ri = (*ref)[c][i];
                        if (ri == ~0u || r0 == ~0u) {
                            wide_rd = wide_rd && ri == r0; // (unread outputs: all three or none)
                            continue;
                        }
                        wide_rd = wide_rd && node_coord[ri / 8u] == i && node_rank[ri / 8u] == node_rank[r0 / 8u] && ri % 8u == r0 % 8u;
                    }
                }
            }
            const auto ref_slot = [&](std::uint32_t ref) { return op_slot(node_coord[ref / 8u], node_rank[ref / 8u], ref % 8u); };
            const auto addr_lists = [&]() {
                std::vector<lds_op> ops;
                // Position reads of the pair lanes: per side a ds_read_b128 (x, y) and a ds_read_b64 (z).
                for (std::uint32_t sd = 0; sd < (vexch ? 0u : 2u); ++sd) {
                    lds_op o16{16, std::vector<std::uint32_t>(pl.L)}, o8{8, std::vector<std::uint32_t>(pl.L)};
                    for (std::uint32_t l = 0; l < pl.L; ++l) {
                        const auto c = l < nc ? l : 0u;
                        std::array<std::uint32_t, 3> tr{};
                        for (std::uint32_t i = 0; i < 3u; ++i) {
                            tr[i] = pl.ext_u[c][pp.de[i][sd]];
                        }
                        const auto b = static_cast<std::uint32_t>(std::find(bodies.begin(), bodies.end(), tr) - bodies.begin());
                        o16.addr[l] = pos_base + 4u * b;
                        o8.addr[l] = pos_base + 4u * b + 2u;
                    }
                    ops.push_back(std::move(o16));
                    ops.push_back(std::move(o8));
                }
                // Operand reads of the glue rounds.
                const auto n_nodes = static_cast<std::uint32_t>(grp.nodes.size());
                for (std::uint32_t r = 0; r * pl.L < n_nodes; ++r) {
                    for (std::uint32_t a = 0; a < n_args; a += 2u) {
                        lds_op o{a + 1u < n_args ? 16 : 8, std::vector<std::uint32_t>(pl.L)};
                        for (std::uint32_t l = 0; l < pl.L; ++l) {
                            const auto j = r * pl.L + l < n_nodes ? r * pl.L + l : r * pl.L;
                            o.addr[l] = op_slot(node_coord[j], node_rank[j], a);
                        }
                        ops.push_back(std::move(o));
                    }
                    // The position coefficients which the round publishes.
                    if (vexch) {
                        continue;
                    }
                    lds_op ow{-8, std::vector<std::uint32_t>(pl.L)};
                    for (std::uint32_t l = 0; l < pl.L; ++l) {
                        if (r * pl.L + l >= n_nodes) {
                            ow.addr[l] = pos_sz + 2u * Dd + blk_used; // (the dummy area)
                            continue;
                        }
                        const auto j = r * pl.L + l;
                        // (The position variable attached to the node: second member of its chain.)
                        const auto &ch = att.at(grp.nodes[j]);
                        std::uint32_t slot = 0;
                        for (const auto var : ch) {
                            for (std::uint32_t b = 0; b < nb; ++b) {
                                for (std::uint32_t i = 0; i < 3u; ++i) {
                                    if (bodies[b][i] == var) {
                                        slot = pos_base + 4u * b + i;
                                    }
                                }
                            }
                        }
                        ow.addr[l] = slot;
                    }
                    ops.push_back(std::move(ow));
                }
                // Stores of the products and of the reactions.
                for (int kind = 0; kind < 2; ++kind) {
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        lds_op o{-8, std::vector<std::uint32_t>(pl.L)};
                        for (std::uint32_t l = 0; l < pl.L; ++l) {
                            const auto ref = l < nc ? (kind == 0 ? pr_ref : rx_ref)[l][i] : ~0u;
                            o.addr[l] = ref != ~0u ? ref_slot(ref) : (kind == 0 ? dummy_pr(i) : dummy_rx(i));
                        }
                        ops.push_back(std::move(o));
                    }
                }
                return ops;
            };
            // Bank model (MI355X_MICROARCH.md, LDS): lane groups per instruction width, banks of 4 bytes; identical addresses
            // broadcast, every further distinct address on a busy bank costs the group one more LDS cycle.
            static const std::vector<std::vector<std::uint32_t>> grp128 = [] {
                std::vector<std::vector<std::uint32_t>> g(4);
                const std::uint32_t r0[] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
                const std::uint32_t r1[] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
                for (const auto x : r0) {
                    g[0].push_back(x);
                    g[2].push_back(x + 32u);
                }
                for (const auto x : r1) {
                    g[1].push_back(x);
                    g[3].push_back(x + 32u);
                }
                return g;
            }();
            const auto wcost = [&](const std::vector<lds_op> &ops, std::uint32_t stride) {
                std::uint64_t tot = 0;
                const auto spw_ = 64u / pl.L;
                const auto lane_addr = [&](const lds_op &o, std::uint32_t lane) {
                    return ((lane / pl.L) % spw_) * stride + o.addr[lane % pl.L];
                };
                for (const auto &o : ops) {
                    const auto n_banks = o.width < 0 ? 32u : 64u;
                    const auto dwords = o.width == 16 ? 4u : 2u;
                    std::vector<std::vector<std::uint32_t>> groups;
                    if (o.width == 16) {
                        groups = grp128;
                    } else if (o.width == 8) {
                        groups.assign(2, {});
                        for (std::uint32_t l = 0; l < 64u; ++l) {
                            groups[l / 32u].push_back(l);
                        }
                    } else {
                        groups.assign(4, {});
                        for (std::uint32_t l = 0; l < 64u; ++l) {
                            groups[l / 16u].push_back(l);
                        }
                    }
                    for (const auto &g : groups) {
                        std::map<std::uint32_t, std::set<std::uint32_t>> banks;
                        for (const auto lane : g) {
                            const auto a = lane_addr(o, lane);
                            for (std::uint32_t d = 0; d < dwords; ++d) {
                                banks[(2u * a + d) % n_banks].insert(a);
                            }
                        }
                        std::size_t mx = 1;
                        for (const auto &[b, st_] : banks) {
                            (void)b;
                            mx = std::max(mx, st_.size());
                        }
                        tot += (mx - 1u) * (o.width < 0 ? 2u : 1u);
                    }
                }
                return tot;
            };
            if (wide_rd) {
                // Placement of the operand arrays inside a coordinate block. A ds_write_b64 is serviced in four groups of 16
                // lanes - one system each - over 16 pairs of banks: the 15 pair lanes + 1 idle lane of a system write 16
                // slots, and the store is conflict free iff those slots are distinct modulo 16. With arrays [t0 .. t4, -] back
                // to back no order of the bodies achieves that for the products AND the reactions (the operand positions a
                // sum reads as reactions are the first ones, as products the last ones), and 34 % of the LDS cycles of the
                // kernel were bank conflicts (profiles/r05_outer_ss_sq_counters.json, first collection). The arrays keep
                // the two 16-byte pairs (t0, t1), (t2, t3) adjacent - read with two ds_read_b128 from ONE table register - and
                // let the fifth operand sit anywhere (a second table register): a depth-first search over (array base, place
                // of the fifth operand) per rank finds a placement in which the slots of every product store and of every
                // reaction store are distinct modulo 16 (outer Solar System: a block of 34 slots).
                const auto BLK = 8u * n_rank;
                bool placed = false;
                if (!v5_flag("nostoreplace") && n_args >= 2u && n_args <= 5u) {
                    // Which operand positions of a rank are written by the product store / the reaction store.
                    std::vector<std::vector<char>> is_rx(n_rank, std::vector<char>(n_args, 0));
                    std::vector<std::vector<char>> is_wr(n_rank, std::vector<char>(n_args, 0));
                    for (std::uint32_t c = 0; c < nc; ++c) {
                        for (int kind = 0; kind < 2; ++kind) {
                            const auto ref = (kind == 0 ? pr_ref : rx_ref)[c][0];
                            if (ref != ~0u) {
                                is_rx[node_rank[ref / 8u]][ref % 8u] = static_cast<char>(kind);
                                is_wr[node_rank[ref / 8u]][ref % 8u] = 1;
                            }
                        }
                    }
                    std::vector<std::vector<std::uint32_t>> cur(n_rank, std::vector<std::uint32_t>(n_args));
                    std::vector<char> used(BLK, 0);
                    std::uint64_t n_visit = 0;
                    const std::function<bool(std::uint32_t, std::uint32_t, std::uint32_t)> dfs = [&](std::uint32_t r, std::uint32_t pm,
                                                                                                 std::uint32_t rm) -> bool {
                        if (r == n_rank) {
                            // The stores of the idle lanes: a free slot on the one residue each store leaves free.
                            std::uint32_t dz[2] = {~0u, ~0u};
                            for (std::uint32_t z = 0; z < BLK; ++z) {
                                if (used[z] != 0) {
                                    continue;
                                }
                                if (dz[0] == ~0u && (pm & (1u << (z % 16u))) == 0u) {
                                    dz[0] = z;
                                } else if (dz[1] == ~0u && (rm & (1u << (z % 16u))) == 0u) {
                                    dz[1] = z;
                                }
                            }
                            if (dz[0] == ~0u || dz[1] == ~0u) {
                                return false;
                            }
                            dummy_pos[0] = dz[0];
                            dummy_pos[1] = dz[1];
                            return true;
                        }
                        const auto n_pair = n_args & ~1u; // operands read in 16-byte pairs
                        for (std::uint32_t e_ = 0; e_ + n_pair <= BLK; e_ += 2u) {
                            for (std::uint32_t f_ = 0; f_ < (n_args % 2u == 1u ? BLK : 1u); ++f_) {
                                if (++n_visit > 4000000u) {
                                    return false;
                                }
                                std::vector<std::uint32_t> ps(n_args);
                                bool ok = true;
                                for (std::uint32_t a_ = 0; a_ < n_args; ++a_) {
                                    ps[a_] = a_ < n_pair ? e_ + a_ : f_;
                                    ok = ok && used[ps[a_]] == 0 && !(a_ >= n_pair && f_ >= e_ && f_ < e_ + n_pair);
                                }
                                if (!ok) {
                                    continue;
                                }
                                std::uint32_t pm2 = pm, rm2 = rm;
                                for (std::uint32_t a_ = 0; a_ < n_args && ok; ++a_) {
                                    if (is_wr[r][a_] == 0) {
                                        continue;
                                    }
                                    auto &m_ = is_rx[r][a_] != 0 ? rm2 : pm2;
                                    const auto bit = 1u << (ps[a_] % 16u);
                                    ok = (m_ & bit) == 0u;
                                    m_ |= bit;
                                }
                                if (!ok) {
                                    continue;
                                }
                                for (const auto x : ps) {
                                    used[x] = 1;
                                }
                                cur[r] = ps;
                                if (dfs(r + 1u, pm2, rm2)) {
                                    return true;
                                }
                                for (const auto x : ps) {
                                    used[x] = 0;
                                }
                            }
                        }
                        return false;
                    };
                    if (pl.L == 16u && dfs(0, 0, 0)) {
                        arr_pos = cur;
                        placed = true;
                    }
                }
                if (placed) {
                    blk_used = std::max(dummy_pos[0], dummy_pos[1]) + 1u;
                    for (const auto &v : arr_pos) {
                        for (const auto x : v) {
                            blk_used = std::max(blk_used, x + 1u);
                        }
                    }
                    blk_used = (blk_used + 1u) & ~1u;
                }
                std::uint64_t best_c = ~std::uint64_t(0);
                auto best_perm = perm;
                std::uint32_t best_D = Dd, best_stride = 0;
                // (+ 2: the dummy area behind the arrays - idle lanes of a partially filled glue round publish there.)
                const auto total_for = [&](std::uint32_t D_) { return pos_sz + 2u * D_ + blk_used + 2u + (bk_in_slab ? 16u : 0u); };
                // The distance between the coordinate blocks and between the slabs of two systems: scanned with the bank model
                // (reads in the lane groups of each instruction width; the stores are settled by the placement above).
                for (std::uint32_t D_ = blk_used; D_ <= blk_used + 6u; D_ += 2u) {
                    Dd = D_;
                    const auto ops = addr_lists();
                    const auto tot = total_for(D_);
                    for (std::uint32_t st_ = (tot + 1u) & ~1u; st_ < ((tot + 1u) & ~1u) + 32u; st_ += 2u) {
                        const auto c = wcost(ops, st_);
                        if (c < best_c) {
                            best_c = c;
                            best_D = D_;
                            best_stride = st_;
                        }
                    }
                }
                perm = best_perm;
                Dd = best_D;
                slab_stride_opt = best_stride;
                bank_cost = best_c;
                if (v5_flag("bankdbg")) {
                    const auto ops = addr_lists();
                    std::fprintf(stderr, "wide layout: D = %u, stride = %u, cost = %llu, perm =", Dd, best_stride,
                                 static_cast<unsigned long long>(best_c));
                    for (const auto x : perm) {
                        std::fprintf(stderr, " %u", x);
                    }
                    std::fprintf(stderr, "\n");
                    for (const auto &o : ops) {
                        std::fprintf(stderr, "  width %d cost %llu addr:", o.width,
                                     static_cast<unsigned long long>(wcost(std::vector<lds_op>{o}, best_stride)));
                        for (const auto x : o.addr) {
                            std::fprintf(stderr, " %u", x);
                        }
                        std::fprintf(stderr, "\n");
                    }
                }
                // The slots.
                std::fill(pl.slot_of.begin(), pl.slot_of.end(), -1);
                for (std::uint32_t b = 0; b < nb && !vexch; ++b) {
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        pl.slot_of[bodies[b][i]] = static_cast<int>(pos_base + 4u * b + i);
                    }
                }
                wide_pr.assign(pl.L, {});
                wide_rx.assign(pl.L, {});
                for (std::uint32_t l = 0; l < pl.L; ++l) {
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        const auto rp = l < nc ? pr_ref[l][i] : ~0u, rr = l < nc ? rx_ref[l][i] : ~0u;
                        wide_pr[l][i] = rp != ~0u ? ref_slot(rp) : dummy_pr(i);
                        wide_rx[l][i] = rr != ~0u ? ref_slot(rr) : dummy_rx(i);
                        if (l < nc && rp != ~0u) {
                            pl.slot_of[pl.clusters[l][pp.pr[i]]] = static_cast<int>(wide_pr[l][i]);
                        }
                        if (l < nc && rr != ~0u) {
                            pl.slot_of[pl.clusters[l][static_cast<std::uint32_t>(pp.rx[i])]] = static_cast<int>(wide_rx[l][i]);
                        }
                    }
                }
                pl.n_slots = total_for(Dd) - 2u - (bk_in_slab ? 16u : 0u);
                (void)out_slot;
            }
        }
    }

    // ---- 2. LDS layout: every slot double-buffered by order parity. ----
    const std::uint32_t max_round_outputs = wide_rd ? 2u : std::max<std::uint32_t>(n_out, one_lane ? 6u : 4u);
    const auto dummy_base = pl.n_slots;
    const auto n_slots_tot = pl.n_slots + max_round_outputs;
    // One-lane pair kernel: ONE buffer. A system never spans wavefronts and the LDS instructions of a wavefront complete
    // in order, so a round may overwrite what it has read as long as its reads come first in the instruction stream -
    // which is how the merged schedule is emitted: reads of x^[k] and of the products of order k-1, then the stores of
    // the products of order k and of x^[k+1] (the stores may alias the loads as far as the compiler knows: it keeps
    // their order).
    const auto buf_stride = one_lane ? 0u : n_slots_tot; // doubles between the two parity buffers
    // (One buffer; the stride comes out of the bank-conflict search above.)
    const auto slab_stride = one_lane ? std::max(slab_stride_opt, n_slots_tot + (bk_in_slab ? 16u : 0u))
                                      : ((2u * n_slots_tot) | 1u); // doubles per system

    // ---- 3. Tables. ----
    std::vector<std::vector<std::uint32_t>> utbl;
    std::vector<std::vector<double>> dtbl;
    // NOTE: tables of slab slots and tables of state-variable indices are kept apart (the slot tables are
    // renumbered by the bank-conflict optimiser below).
    std::vector<char> utbl_is_slot;
    // utexpr[t]: how the kernel refers to the per-lane value of table t - a register loaded at the top of the kernel
    // ("ut<t>"), or (one-lane pair kernel, where every register counts) an earlier table plus a constant when the two
    // differ by the same amount on every lane (the three coordinates of a body, the three products of a pair: the
    // constant folds into the offset field of the LDS instruction).
    std::vector<std::string> utexpr;
    const auto add_utbl = [&](std::vector<std::uint32_t> v, bool is_slot = true) {
        // Deduplicate identical tables.
        for (std::size_t t = 0; t < utbl.size(); ++t) {
            if (utbl[t] == v && (utbl_is_slot[t] != 0) == is_slot) {
                return t;
            }
        }
        std::string ex = "ut" + std::to_string(utbl.size());
        for (std::size_t t = 0; one_lane && is_slot && t < utbl.size(); ++t) {
            if (utbl_is_slot[t] == 0 || utexpr[t] != "ut" + std::to_string(t)) {
                continue;
            }
            const auto d = static_cast<std::int64_t>(v[0]) - static_cast<std::int64_t>(utbl[t][0]);
            bool affine = true;
            for (std::size_t l2 = 0; l2 < v.size(); ++l2) {
                affine = affine && (static_cast<std::int64_t>(v[l2]) - static_cast<std::int64_t>(utbl[t][l2]) == d);
            }
            if (affine) {
                ex = "(ut" + std::to_string(t) + (d >= 0 ? " + " : " - ") + std::to_string(d >= 0 ? d : -d) + "u)";
                break;
            }
        }
        utexpr.push_back(std::move(ex));
        utbl.push_back(std::move(v));
        utbl_is_slot.push_back(is_slot ? 1 : 0);
        return utbl.size() - 1u;
    };
    const auto add_dtbl = [&](std::vector<double> v) {
        dtbl.push_back(std::move(v));
        return dtbl.size() - 1u;
    };
    const auto utname = [&](std::size_t t) { return utexpr[t]; };
    // (One-lane pair kernel: the per-lane constants live in LDS and are read where they are used - one address register
    // for all of them instead of two registers each.)
    const auto dtname = [&](std::size_t t) {
        return one_lane ? ("dtl[" + std::to_string(t * L) + "]") : ("dt" + std::to_string(t));
    };
    // (One-lane pair kernel, fused reactions: the coefficients of the sums are loaded once per step - registers through the
    // orders, free again in the tail of the step; "frxlds" among HEYOKA_AMD_V5_OPTS reads them from LDS at every use.)
    const bool frx_regs = frx && !v5_flag("frxlds");
    const auto coefname = [&](std::size_t t) { return frx_regs ? ("frc" + std::to_string(t)) : dtname(t); };

    ssa_emitter e(p, order);
    auto &os = e.os;
    // Lane-pair kernel: x^[k+1] = f^[k] * RN(1 / (k + 1)) - one multiplication, within 1 ulp of the quotient - instead of
    // the exact 3-operation sequence (60 VALU instructions per step, +2.1 % system-steps/s; the strict-contraction parity
    // test passes its 1e4 / 1e5 eps bounds with it). kw::exact_division restores the correctly-rounded quotient.
    e.recip_div = pairk && !opts.exact_division;
    e.enable_pow_rcp(!opts.exact_division);

    // Lane-pair variant: lane l = 2 * pair + role (role 0 = A: d_0, d_1; role 1 = B: d_2 and the pow); the lanes
    // beyond the last pair replicate pair 0 and write to dummy slots.
    struct pair_tables {
        std::size_t s0 = 0, s1 = 0, p0 = 0, p1 = 0, os = 0, op = 0, rs = 0, rp = 0, csc = 0, crs = 0, crp = 0;
    } pt;
    // Slab slot through which a product pr travels (its own, or - when only its reaction was exported - that one's).
    const auto pr_slot = [&](std::uint32_t pr_u, std::uint32_t rx_u, std::uint32_t dflt) {
        if (pl.slot_of[pr_u] >= 0) {
            return static_cast<std::uint32_t>(pl.slot_of[pr_u]);
        }
        return (fuse_rx && pl.slot_of[rx_u] >= 0) ? static_cast<std::uint32_t>(pl.slot_of[rx_u]) : dflt;
    };
    if (pair_split) {
        const auto slot_or = [&](std::uint32_t u, std::uint32_t dflt) {
            return pl.slot_of[u] >= 0 ? static_cast<std::uint32_t>(pl.slot_of[u]) : dflt;
        };
        std::vector<std::uint32_t> s0(L), s1(L), p0(L), p1(L), os_(L), op(L), rs(L), rp(L);
        std::vector<double> csc(L, 1.), crs(L, 0.), crp(L, 0.);
        for (std::uint32_t l = 0; l < L; ++l) {
            const bool valid = l / 2u < nc;
            const auto c = valid ? l / 2u : 0u;
            const bool rb = (l & 1u) != 0u;
            const auto ds = rb ? 2u : 0u;
            const auto &cl = pl.clusters[c];
            const auto ext = [&](std::uint32_t dc, std::uint32_t a) {
                return static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][pp.de[dc][a]]]);
            };
            s0[l] = ext(ds, 0);
            s1[l] = ext(ds, 1);
            p0[l] = rb ? s0[l] : ext(1, 0);
            p1[l] = rb ? s0[l] : ext(1, 1);
            if (fuse_rx) {
                os_[l] = valid ? pr_slot(cl[pp.pr[ds]], cl[static_cast<std::uint32_t>(pp.rx[ds])], dummy_base) : dummy_base;
                op[l] = (valid && !rb) ? pr_slot(cl[pp.pr[1]], cl[static_cast<std::uint32_t>(pp.rx[1])], dummy_base + 1u)
                                       : dummy_base + 1u;
            } else {
                os_[l] = valid ? slot_or(cl[pp.pr[ds]], dummy_base) : dummy_base;
                op[l] = (valid && !rb) ? slot_or(cl[pp.pr[1]], dummy_base + 1u) : dummy_base + 1u;
            }
            if (pp.rx[0] >= 0 && !fuse_rx) {
                rs[l] = valid ? slot_or(cl[static_cast<std::uint32_t>(pp.rx[ds])], dummy_base + 2u) : dummy_base + 2u;
                rp[l] = (valid && !rb) ? slot_or(cl[static_cast<std::uint32_t>(pp.rx[1])], dummy_base + 3u) : dummy_base + 3u;
                crs[l] = p.nodes[cl[static_cast<std::uint32_t>(pp.rx[ds])] - n_eq].args[0].value;
                crp[l] = rb ? 0. : p.nodes[cl[static_cast<std::uint32_t>(pp.rx[1])] - n_eq].args[0].value;
            }
            if (pp.sc >= 0) {
                csc[l] = p.nodes[cl[static_cast<std::uint32_t>(pp.sc)] - n_eq].args[0].value;
            }
        }
        pt.s0 = add_utbl(std::move(s0));
        pt.s1 = add_utbl(std::move(s1));
        pt.p0 = add_utbl(std::move(p0));
        pt.p1 = add_utbl(std::move(p1));
        pt.os = add_utbl(std::move(os_));
        pt.op = add_utbl(std::move(op));
        if (pp.rx[0] >= 0 && !fuse_rx) {
            pt.rs = add_utbl(std::move(rs));
            pt.rp = add_utbl(std::move(rp));
            pt.crs = add_dtbl(std::move(crs));
            pt.crp = add_dtbl(std::move(crp));
        }
        if (pp.sc >= 0) {
            pt.csc = add_dtbl(std::move(csc));
        }
    }
    // One-lane pair kernel: lane l = pair l (the lanes beyond the last pair replicate pair 0 and write to dummy slots).
    struct single_tables {
        std::size_t s[3][2] = {}, o[3] = {}, r[3] = {}, csc = 0, crs = 0;
    } st1;
    if (one_lane) {
        std::vector<double> csc(L, 1.), crs(L, 0.);
        for (std::uint32_t i = 0; i < 3u; ++i) {
            std::vector<std::uint32_t> s0(L), s1(L), o(L), r(L);
            for (std::uint32_t l = 0; l < L; ++l) {
                const bool valid = l < nc;
                const auto c = valid ? l : 0u;
                const auto &cl = pl.clusters[c];
                if (vexch) {
                    // (Velocity exchange: the jet columns of the two bodies instead of slab slots.)
                    s0[l] = vx_col[l][0] + i;
                    s1[l] = vx_col[l][1] + i;
                } else {
                    s0[l] = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][pp.de[i][0]]]);
                    s1[l] = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][pp.de[i][1]]]);
                }
                // (Every lane owns its output slots, the idle ones too: slot = first slot of pair 0 + 3 * lane + i.)
                o[l] = wide_rd ? wide_pr[l][i]
                               : static_cast<std::uint32_t>(pl.slot_of[pl.clusters[0][pp.pr[i]]]) - 3u * lane_pr[0] + 3u * lane_pr[l];
                if (pp.rx[0] >= 0 && !fuse_rx) {
                    const auto ru = cl[static_cast<std::uint32_t>(pp.rx[i])];
                    r[l] = wide_rd ? wide_rx[l][i]
                                   : static_cast<std::uint32_t>(pl.slot_of[pl.clusters[0][static_cast<std::uint32_t>(pp.rx[i])]])
                                         - 3u * lane_rx[0] + 3u * lane_rx[l];
                    const auto cv = p.nodes[ru - n_eq].args[0].value;
                    if (i > 0u && cv != crs[l]) {
                        why_not = "one-lane pair kernel: the reaction coefficients of a pair differ between the coordinates";
                        return ret;
                    }
                    crs[l] = cv;
                }
                if (pp.sc >= 0) {
                    csc[l] = p.nodes[cl[static_cast<std::uint32_t>(pp.sc)] - n_eq].args[0].value;
                }
            }
            st1.s[i][0] = add_utbl(std::move(s0));
            st1.s[i][1] = add_utbl(std::move(s1));
            st1.o[i] = add_utbl(std::move(o));
            if (pp.rx[0] >= 0 && !fuse_rx) {
                st1.r[i] = add_utbl(std::move(r));
            }
        }
        if (pp.sc >= 0) {
            st1.csc = add_dtbl(std::move(csc));
        }
        if (pp.rx[0] >= 0 && !fuse_rx) {
            st1.crs = add_dtbl(std::move(crs));
        }
    }
    std::vector<std::size_t> ext_tbl(n_ext), out_tbl(n_out), cst_tbl(n_cst);
    for (std::uint32_t x = 0; !pairk && x < n_ext; ++x) {
        std::vector<std::uint32_t> v(L);
        for (std::uint32_t l = 0; l < L; ++l) {
            v[l] = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[l < nc ? l : 0u][x]]);
        }
        ext_tbl[x] = add_utbl(std::move(v));
    }
    for (std::uint32_t x = 0; !pairk && x < n_out; ++x) {
        std::vector<std::uint32_t> v(L);
        for (std::uint32_t l = 0; l < L; ++l) {
            v[l] = l < nc ? static_cast<std::uint32_t>(pl.slot_of[pl.clusters[l][pl.out_pos[x]]]) : dummy_base + x;
        }
        out_tbl[x] = add_utbl(std::move(v));
    }
    for (std::uint32_t x = 0; !pairk && x < n_cst; ++x) {
        std::vector<double> v(L);
        for (std::uint32_t l = 0; l < L; ++l) {
            v[l] = pl.cst_val[l < nc ? l : 0u][x];
        }
        cst_tbl[x] = add_dtbl(std::move(v));
        const auto [q, a] = pl.cst_pos[x];
        e.numpar_override[&p.nodes[t0[q] - n_eq].args[a]] = "ccst" + std::to_string(x);
    }
    // Per-lane parameters: tables of parameter indices; the values are loaded when a group of systems is picked up.
    std::vector<std::size_t> lane_par_tbls;
    const auto lane_par = [&](std::vector<std::uint32_t> idx) {
        const auto t = add_utbl(std::move(idx), false);
        if (std::find(lane_par_tbls.begin(), lane_par_tbls.end(), t) == lane_par_tbls.end()) {
            lane_par_tbls.push_back(t);
        }
        return "lp" + std::to_string(t);
    };
    for (std::size_t x = 0; !pairk && x < pl.par_pos.size(); ++x) {
        std::vector<std::uint32_t> v(L);
        for (std::uint32_t l = 0; l < L; ++l) {
            v[l] = pl.par_idx[l < nc ? l : 0u][x];
        }
        const auto [q, a] = pl.par_pos[x];
        e.numpar_override[&p.nodes[t0[q] - n_eq].args[a]] = lane_par(std::move(v));
    }

    // Glue rounds (+ the owner slots of the attached state variables).
    struct owner_slot {
        std::size_t out_tbl = 0;  // slab slot of the state variable
        std::size_t var_tbl = 0;  // state-variable index (for the global state array)
        std::uint32_t col = 0;    // owner slot id
        std::uint32_t cbase = 0;  // first jet column of the slot (columns are compressed: one per valid lane)
        std::uint32_t n_valid = 0;
        std::vector<std::string> xname; // SSA names of the coefficients, by order
        bool slab_needed = true;        // is one of the variables of the slot read through the slab?
        // One-lane pair kernel: the second variable of a chain (x' = v) keeps no jet column: its coefficients are
        // re-derived from the column of the first one (parent) in the final evaluation; cbase then counts the
        // order-0 entries of the derived variables (their current values).
        bool derived = false;
        std::uint32_t parent = 0; // owner slot id of the variable it is derived from
    };
    struct glue_round {
        std::vector<std::size_t> arg_tbl;
        std::size_t out_tbl = 0;
        std::uint32_t n_valid = 0; // lanes l < n_valid own a real node
        bool exported = true;
        std::vector<owner_slot> owners;
        std::vector<std::string> par_name; // per-lane parameter value names, by argument (empty: none)
        std::vector<std::string> c0name;   // names of the constant operands read at order 0, by argument
        std::vector<std::size_t> coef_tbl; // reaction fusion: per-lane coefficient tables, by argument (empty: not fused)
    };
    std::vector<std::vector<glue_round>> rounds(pl.groups.size());
    std::uint32_t n_own = 0, n_col_acc = 0, n_dcol_acc = 0;
    for (std::size_t g = 0; g < pl.groups.size(); ++g) {
        const auto &grp = pl.groups[g];
        const auto n_nodes = static_cast<std::uint32_t>(grp.nodes.size());
        const auto n_rounds = (n_nodes + L - 1u) / L;
        const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
        for (std::uint32_t r = 0; r < n_rounds; ++r) {
            glue_round gr;
            gr.n_valid = std::min(L, n_nodes - r * L);
            const auto node_of = [&](std::uint32_t l) {
                const auto j = r * L + l;
                return grp.nodes[j < n_nodes ? j : r * L];
            };
            bool round_fused = false;
            for (std::uint32_t l = 0; fuse_rx && l < L; ++l) {
                for (const auto &o : p.nodes[node_of(l) - n_eq].args) {
                    round_fused = round_fused || (is_var(o) && rx_fused[o.idx] != 0);
                }
            }
            for (std::size_t a = 0; a < n0.args.size(); ++a) {
                if (is_var(n0.args[a]) && round_fused) {
                    std::vector<std::uint32_t> v(L);
                    std::vector<double> cf(L, 1.);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        const auto ua = p.nodes[node_of(l) - n_eq].args[a].idx;
                        if (rx_fused[ua] != 0) {
                            v[l] = pr_slot(rx_src[ua], ua, 0);
                            cf[l] = p.nodes[ua - n_eq].args[0].value;
                        } else {
                            v[l] = static_cast<std::uint32_t>(pl.slot_of[ua]);
                        }
                    }
                    gr.arg_tbl.push_back(add_utbl(std::move(v)));
                    gr.coef_tbl.push_back(add_dtbl(std::move(cf)));
                } else if (is_var(n0.args[a])) {
                    std::vector<std::uint32_t> v(L);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        v[l] = static_cast<std::uint32_t>(pl.slot_of[p.nodes[node_of(l) - n_eq].args[a].idx]);
                    }
                    gr.arg_tbl.push_back(add_utbl(std::move(v)));
                } else if (n0.args[a].type == operand::kind::num) {
                    std::vector<double> v(L);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        v[l] = p.nodes[node_of(l) - n_eq].args[a].value;
                    }
                    gr.arg_tbl.push_back(add_dtbl(std::move(v)));
                } else {
                    // Parameter: per-lane index (the nodes of a group may read different parameters).
                    std::vector<std::uint32_t> v(L);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        v[l] = p.nodes[node_of(l) - n_eq].args[a].idx;
                    }
                    gr.par_name.resize(n0.args.size());
                    gr.par_name[a] = lane_par(std::move(v));
                    gr.arg_tbl.push_back(0);
                }
            }
            bool any_read = false;
            std::vector<std::uint32_t> v(L);
            for (std::uint32_t l = 0; l < L; ++l) {
                const auto j = r * L + l;
                v[l] = (j < n_nodes && pl.slot_of[grp.nodes[j]] >= 0) ? static_cast<std::uint32_t>(pl.slot_of[grp.nodes[j]]) : dummy_base;
                any_read = any_read || (j < n_nodes && glue_read[grp.nodes[j]] != 0);
            }
            gr.exported = any_read;
            gr.out_tbl = add_utbl(std::move(v));
            for (std::uint32_t a = 0; a < grp_natt[g]; ++a) {
                owner_slot ow;
                std::vector<std::uint32_t> vs(L), vv(L);
                for (std::uint32_t l = 0; l < L; ++l) {
                    const auto var = att.at(node_of(l))[a];
                    vs[l] = (r * L + l < n_nodes && pl.slot_of[var] >= 0) ? static_cast<std::uint32_t>(pl.slot_of[var]) : dummy_base;
                    vv[l] = var;
                }
                ow.out_tbl = add_utbl(std::move(vs));
                ow.var_tbl = add_utbl(std::move(vv), false);
                ow.col = n_own++;
                ow.n_valid = gr.n_valid;
                if (one_lane && a >= 1u) {
                    ow.derived = true;
                    ow.parent = gr.owners[a - 1u].col;
                    ow.cbase = n_dcol_acc;
                    n_dcol_acc += gr.n_valid;
                } else {
                    ow.cbase = n_col_acc;
                    n_col_acc += gr.n_valid;
                }
                ow.xname.resize(order + 1u);
                ow.slab_needed = false;
                for (std::uint32_t l = 0; l < L && r * L + l < n_nodes; ++l) {
                    ow.slab_needed = ow.slab_needed || glue_read[att.at(grp.nodes[r * L + l])[a]] != 0;
                }
                // (Velocity exchange: nobody reads a position coefficient - the compiler drops the unused ones.)
                ow.slab_needed = ow.slab_needed && !vexch;
                gr.owners.push_back(std::move(ow));
            }
            rounds[g].push_back(std::move(gr));
        }
    }
    if (n_own == 0u) {
        why_not = "no state variable could be attached to a glue round";
        return ret;
    }
    // Columns of the state-variable jets: one per state variable (owner slots are compressed: only the valid lanes
    // of a slot own a column) plus one dummy column per system which absorbs the stores of the idle lanes of a
    // partially filled slot - they replicate the node of a valid lane, so every statement of the step body is
    // unconditional (no exec-mask manipulation inside the step loop).
    const auto n_col = n_col_acc;
    // (One-lane pair kernel: no dummy column - the idle lanes of a partially filled slot store to the entry they
    // replicate - and a row is laid out [owner slot][system][lane]: the 32 lanes which a ds_read_b64 services together
    // (two systems) then touch 32 consecutive doubles, i.e. every bank once.)
    const auto n_colp = one_lane ? n_col : n_col + 1u;
    // (One-lane pair kernel: current values of the derived variables, [system of the wave][entry] + one dummy entry.)
    const auto n_dcol = n_dcol_acc;
    const auto n_dcolp = one_lane ? n_dcol : n_dcol + 1u;
    const auto n_hslots = (n_col + L - 1u) / L; // lane slots of the final Horner / compensated evaluation
    // Position of an owner slot inside a row of the jets (one-lane pair kernel): rows are [owner slot][system][lane] - the 32
    // lanes which a ds_read_b64 services together touch 32 consecutive doubles - or, with the velocity exchange,
    // [system][column] - the three coordinates of a body adjacent. jet_off: first entry of the slot for the first system
    // of the wavefront, jet_sys: distance between two systems.
    const auto jet_off = [&](const owner_slot &ow) -> std::uint64_t {
        return vexch ? ow.cbase : static_cast<std::uint64_t>(spw) * ow.cbase;
    };
    const auto jet_sys = [&](const owner_slot &ow) -> std::uint32_t { return vexch ? (ow.derived ? n_dcol : n_col) : ow.n_valid; };
    // Jets of the state variables: [order][system of the wave][column], per wave. Kept in LDS when the
    // block's slab + jets fit in the 160 KB of a CU (the kernel occupies a whole CU anyway: 512 registers
    // per lane), otherwise in a per-wave global scratch.
    const auto jet_rows_doubles = static_cast<std::uint64_t>(order + 1u) * spw * n_colp;
    const auto jet_doubles_per_wave = jet_rows_doubles + (one_lane ? static_cast<std::uint64_t>(spw) * n_dcolp : 0u);
    const auto lds_doubles_slab = static_cast<std::uint64_t>(wpb) * spw * slab_stride;
    // (Mode 4: plus the source table of the cooperative store of the Taylor coefficients, 4 bytes per row.)
    const auto lds_tc_table_bytes = m4 ? static_cast<std::uint64_t>(n_eq) * (order + 1u) * 8u : 0u;
    const bool jet_lds = (lds_doubles_slab + wpb * jet_doubles_per_wave) * 8u + lds_tc_table_bytes <= 160u * 1024u;
    if (vexch && (!jet_lds || n_dcol != n_col)) {
        why_not = "one-lane pair kernel: the velocity exchange needs the jets in LDS and one position per velocity";
        return ret;
    }
    // Stepper with events: compact set of Taylor coefficients (see emitted_module::compact_tc) through the cooperative
    // store of the LDS-resident jets. HEYOKA_AMD_COMPACT_TC=0 switches it off (A/B measurements).
    std::size_t n_tc_rows = 0; // rows of the mode-4 store (set where its source table is emitted)
    const bool compact_tc = [&]() {
        if (!m4 || !jet_lds) {
            return false;
        }
        if (one_lane) {
            return true;
        }
        if (!opts.dev.compact_tc) {
            return false;
        }
        // (Chains x' = v, v' = a only: the parent of a derived variable is not derived itself.)
        const auto derived = [&](std::uint32_t var) {
            return p.sv_defs[var].type == operand::kind::uvar && p.sv_defs[var].idx < n_eq;
        };
        bool any = false;
        for (std::uint32_t var = 0; var < n_eq; ++var) {
            if (derived(var)) {
                if (derived(p.sv_defs[var].idx)) {
                    return false;
                }
                any = true;
            }
        }
        return any;
    }();

    // Final evaluation of a partially filled owner slot with a derived variable (x' = v; 18 velocity columns on 16 lanes
    // leave 2): ONE series per lane - the lanes [0, n) sum the velocity columns, the lanes [n, 2 n) the series derived from
    // them, whose coefficient k is row k - 1 of the same column times RN(1 / k). Both kinds run the same statements: where
    // the lane's current value lives, where row k of its series starts and which row of the factor table (ones / RN(1 / k))
    // it reads are per-lane table entries (pk_tbl[owner slot] = the three tables).
    const auto pack_tail_slot = [&](const owner_slot &ow, const owner_slot *dv) {
        return one_lane && jet_lds && !m4 && opts.high_accuracy && dv != nullptr && !ow.derived && 2u * ow.n_valid <= L
               && !v5_flag("nopack2");
    };
    std::map<std::uint32_t, std::array<std::size_t, 3>> pk_tbl;
    for (const auto &rg : rounds) {
        for (const auto &gr : rg) {
            for (const auto &ow : gr.owners) {
                for (const auto &o2 : gr.owners) {
                    if (!(o2.derived && o2.parent == ow.col && pack_tail_slot(ow, &o2))) {
                        continue;
                    }
                    const auto nv = ow.n_valid;
                    const auto kst = static_cast<std::uint64_t>(spw) * n_colp;
                    std::vector<std::uint32_t> tv(L), tj(L), tf(L);
                    for (std::uint32_t l = 0; l < L; ++l) {
                        const bool isx = l >= nv && l < 2u * nv;
                        const auto c = isx ? l - nv : (l < nv ? l : 0u);
                        tv[l] = static_cast<std::uint32_t>((isx ? jet_rows_doubles + jet_off(o2) : jet_off(ow)) + c);
                        tj[l] = static_cast<std::uint32_t>(jet_off(ow) + c + (isx ? 0u : kst));
                        tf[l] = isx ? order + 1u : 0u;
                    }
                    pk_tbl[ow.col] = {add_utbl(std::move(tv), false), add_utbl(std::move(tj), false), add_utbl(std::move(tf), false)};
                }
            }
        }
    }

    // Merged schedule (lane-pair variant, one glue level after the clusters): round k = cluster(k) + glue(k-1),
    // one LDS synchronisation per order instead of two.
    const bool merged = [&]() {
        if (!pairk || pl.cluster_level != 1u || pl.max_level != 2u) {
            return false;
        }
        for (const auto &g : pl.groups) {
            if (g.level != 2u) {
                return false;
            }
        }
        return true;
    }();
    if (one_lane && !merged) {
        why_not = "one-lane pair kernel: the merged schedule does not apply";
        return ret;
    }

    // ---- 4. Emission helpers. ----
    const auto slabk = [&](std::uint32_t k, const std::string &tbl) {
        // Parity buffer of order k.
        return (k % 2u == 0u || buf_stride == 0u) ? ("slab[" + tbl + "]")
                                                  : ("slab[" + tbl + " + " + std::to_string(buf_stride) + "u]");
    };
    const auto jet_at = [&](std::uint32_t k, std::uint32_t col) {
        return "jc" + std::to_string(col) + "[" + std::to_string(static_cast<std::uint64_t>(k) * spw * n_colp) + "]";
    };
    const auto sync = [&]() { os << "HY_WSYNC();\n"; };
    // Where the current value (order 0) of the variables of an owner slot lives: read side (the idle lanes of a partially
    // filled slot read the entry of a valid lane) and write side (... and write to the dummy entry).
    const auto row0_r = [&](const owner_slot &ow) {
        return (ow.derived ? "x0r" : "jr") + std::to_string(ow.col) + "[0]";
    };
    const auto row0_w = [&](const owner_slot &ow) {
        return (ow.derived ? "x0c" : "jc") + std::to_string(ow.col) + "[0]";
    };

    // Owner-slot bookkeeping when a new coefficient of a state variable is produced.
    const auto publish_sv = [&](owner_slot &ow, std::uint32_t k, const std::string &name) {
        ow.xname[k] = name;
        if (ow.slab_needed) {
            os << slabk(k, utname(ow.out_tbl)) << " = " << name << ";\n";
        }
        if (k != 0u && !ow.derived) {
            // (The order-0 row of the jets *is* the current state: written by the update of the previous step.)
            os << jet_at(k, ow.col) << " = " << name << ";\n";
        }
        // NOTE: the idle lanes of a partially filled slot hold a copy of a valid lane's coefficient: harmless in a maximum.
        const char *acc = (k == 0u) ? "m0" : (k == order ? "mo" : (k == order - 1u ? "mom1" : nullptr));
        if (acc != nullptr) {
            os << acc << " = hy_nmax(" << acc << ", fabs(" << name << "));\n";
        }
    };

    // A 16-byte LDS read of two adjacent slots (wide-read layout: the table entries are even, the slab is 16-byte aligned).
    std::uint32_t n_wide = 0;
    const auto wide_read = [&](const std::string &tbl) {
        const auto nm = "w" + std::to_string(n_wide++);
        os << "const hy_d2 " << nm << " = *reinterpret_cast<const hy_d2 *>(slab + " << tbl << ");\n";
        ++e.n_stmt;
        return nm;
    };
    // A glue round is emitted in two halves: the LDS reads of the operands, and the computation (node rule,
    // export, fused state-variable recursions). In overlap mode independent FMA work is placed in between.
    const auto emit_glue_reads = [&](std::size_t g, std::uint32_t r, std::uint32_t k) {
        const auto &grp = pl.groups[g];
        auto &gr = rounds[g][r];
        const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
        std::vector<std::string> names(n0.args.size());
        for (std::size_t a = 0; a < n0.args.size(); ++a) {
            if (wide_rd && a + 1u < n0.args.size() && a % 2u == 0u) {
                // (Operands a, a + 1 of the sum: adjacent slots of the node's operand array.)
                const auto w = wide_read(utname(gr.arg_tbl[a]));
                names[a] = w + ".x";
                names[a + 1u] = w + ".y";
                ++a;
                continue;
            }
            if (is_var(n0.args[a])) {
                names[a] = e.def(slabk(k, utname(gr.arg_tbl[a])));
            }
        }
        return names;
    };
    const auto emit_glue_compute = [&](std::size_t g, std::uint32_t r, std::uint32_t k,
                                       const std::vector<std::string> &names) {
        const auto &grp = pl.groups[g];
        auto &gr = rounds[g][r];
        const auto rep = grp.nodes[0];
        const auto &n0 = p.nodes[rep - n_eq];
        const auto saved = e.numpar_override;
        std::vector<std::pair<std::uint32_t, std::string>> saved_vals, saved_vals0;
        std::string fused_val;
        if (!gr.coef_tbl.empty()) {
            // Sum of scaled products, pairwise like the sum rule: ((t0 + t1) + (t2 + t3)) + ..., t_i = c_i * p_i, the
            // first product of every pair fused into the addition.
            std::vector<std::string> terms;
            for (std::size_t a = 0; a + 1u < names.size(); a += 2u) {
                const auto m = e.def(ssa_emitter::mul(coefname(gr.coef_tbl[a + 1u]), names[a + 1u]));
                terms.push_back(e.def("__builtin_fma(" + coefname(gr.coef_tbl[a]) + ", " + names[a] + ", " + m + ")"));
            }
            const bool odd = names.size() % 2u == 1u;
            while (terms.size() > 1u) {
                std::vector<std::string> nt;
                for (std::size_t i = 0; i + 1u < terms.size(); i += 2u) {
                    nt.push_back(e.def(terms[i] + " + " + terms[i + 1u]));
                }
                if (terms.size() % 2u == 1u) {
                    nt.push_back(terms.back());
                }
                terms = std::move(nt);
            }
            if (odd) {
                const auto a = names.size() - 1u;
                fused_val = terms.empty() ? e.def(ssa_emitter::mul(coefname(gr.coef_tbl[a]), names[a]))
                                          : e.def("__builtin_fma(" + coefname(gr.coef_tbl[a]) + ", " + names[a] + ", " + terms[0] + ")");
            } else {
                fused_val = terms[0];
            }
        }
        for (std::size_t a = 0; fused_val.empty() && a < n0.args.size(); ++a) {
            const auto &o = n0.args[a];
            if (is_var(o)) {
                saved_vals.emplace_back(o.idx, e.val(o.idx, k));
                e.val(o.idx, k) = names[a];
                // Constant operand: the linear rule of ssa_emitter::node() uses the value read at order 0.
                if (cu[o.idx] != 0) {
                    if (k == 0u) {
                        gr.c0name.resize(n0.args.size());
                        gr.c0name[a] = names[a];
                    } else {
                        saved_vals0.emplace_back(o.idx, e.val(o.idx, 0));
                        e.val(o.idx, 0) = gr.c0name.at(a);
                    }
                }
            } else if (o.type == operand::kind::num) {
                e.numpar_override[&o] = dtname(gr.arg_tbl[a]);
            } else if (a < gr.par_name.size() && !gr.par_name[a].empty()) {
                e.numpar_override[&o] = gr.par_name[a];
            }
        }
        if (n0.kind == func_kind::prod && n0.args[0].type == operand::kind::num && n0.args[0].value == -1.) {
            e.numpar_override.erase(&n0.args[0]);
        }
        if (fused_val.empty()) {
            e.node(rep - n_eq, k);
        }
        const auto gval = fused_val.empty() ? e.val(rep, k) : fused_val;
        // NOTE: constant nodes are exported at every order too (zeros beyond order 0): a reader whose template position
        // pairs the constant with a variable in another cluster reads it at every order.
        if (gr.exported) {
            os << slabk(k, utname(gr.out_tbl)) << " = " << gval << ";\n";
        }
        for (auto it = saved_vals.rbegin(); it != saved_vals.rend(); ++it) {
            e.val(it->first, k) = it->second;
        }
        for (auto it = saved_vals0.rbegin(); it != saved_vals0.rend(); ++it) {
            e.val(it->first, 0) = it->second;
        }
        e.numpar_override = saved;

        // Fused state-variable recursion: x^[k+1] = src^[k] / (k + 1). In the merged schedule the a-th variable of
        // the chain runs a orders ahead (x^[k+1+a] from the coefficient of order k + a of its predecessor, which
        // the same lane has just produced): a position is then known one exchange earlier than the acceleration
        // of the same order, which is what lets cluster(k+1) and glue(k) share one round.
        for (std::size_t a = 0; a < gr.owners.size(); ++a) {
            const auto ord = k + 1u + (merged ? static_cast<std::uint32_t>(a) : 0u);
            if (ord > order) {
                continue;
            }
            const auto src = (a == 0u) ? gval : gr.owners[a - 1u].xname[ord - 1u];
            const auto x = e.div_const(src, ord);
            publish_sv(gr.owners[a], ord, x);
        }
    };
    const auto emit_glue_round = [&](std::size_t g, std::uint32_t r, std::uint32_t k) {
        emit_glue_compute(g, r, k, emit_glue_reads(g, r, k));
    };

    // NOTE: the history chains of order k can be emitted in several parts: the first one at the end of
    // order k - 1 (it overlaps the glue exchange), the others at the beginning of the cluster phase of
    // order k. Measured on gfx950 (outer-SS, 1 048 576 systems): 18.11 ms per launch for 1, 2 and 3 parts
    // - the chains only touch registers, so the compiler's scheduler already moves them across the
    // compiler-only HY_WSYNC barrier. Hence the default of a single part.
    const std::uint32_t n_parts = 1;
    std::vector<std::uint32_t> t0_ids;
    for (const auto u : t0) {
        t0_ids.push_back(u - n_eq);
    }

    // Explicit overlap of the LDS exchange latency (the workgroup runs one wavefront per SIMD, nobody else
    // hides it): in both exchange regions of an order the LDS reads are issued first, then - fenced by
    // scheduling barriers - a chunk of history-chain FMAs which do not depend on them, then the dependent
    // computation. The chunks are (ssa_emitter::emit_partials_sel): in the cluster region of order k the second
    // half of the early terms of order k + 1; in the last glue region of order k the late terms of order k + 1
    // and the first half of the early terms of order k + 2.
    const bool overlap = true, fence2 = true;
    const auto sched_fence = [&]() { os << "__builtin_amdgcn_sched_barrier(0);\n"; };
    using psel = ssa_emitter::part_sel;

    // ---- Lane-pair cluster program ----
    // Coefficient histories of a lane (SSA names by order), role A | role B:
    //   aS: d_0 | d_2          aP: d_1 | b = sum of squares          aR: sa = (scaled) pow, both lanes
    //   aRp: d_1 (a copy) | -(alpha + 1) j sa_j
    // Convolution chains of order k (same FMA stream on both lanes), history part = indices 1 .. k-1:
    //   c1 = sum aP[k-j] aR[j]   (A: d_1 * sa,  B: S1 = sum b[k-j] sa[j] of the pow recurrence)
    //   c2 = sum aP[k-j] aRp[j]  (A: the order-k coefficient of d_1^2, B: S2 = sum b[k-j] j sa[j])
    //   c3 = sum aS[k-j] aR[j]   (d_0 * sa | d_2 * sa)
    //   c4 = sum_{j <= jmax} aS[k-j] aS[j] (+ the middle square): d_0^2 | d_2^2
    // The pow recurrence (src/math/pow.cpp:517-549) k b_0 a_k = sum_{j<k} (k alpha - j (alpha + 1)) b_{k-j} a_j is linear
    // in a: it is run directly on sa = c a, as alpha k S1 - (alpha + 1) S2.
    // Per order the two lanes exchange (DPP quad_perm [1,0,3,2], no LDS): the partial sums of squares, then sa_k.
    std::vector<std::string> aP(order + 1u), aR(order + 1u), aRp(order + 1u), aS(order + 1u);
    std::string hc1, hc2, hc3, hc4, hmid;
    const bool has_rx = pp.rx[0] >= 0;
    std::string rb1;   // 1 / b_0 (lane B)
    // Normalised pow recurrence (default): lane B keeps b_k / b_0 (k >= 1) instead of b_k, one multiplication by
    // RN(1 / b_0) folded into the FMA which forms the lane's aP[k]; the recurrence k b_0 a_k = sum(...) then needs no
    // division: a_k = alpha S1 - (alpha + 1) S2 / k on the normalised sums. The new coefficient goes to both lanes of
    // the pair with one DPP broadcast from the odd lane. kw::exact_division: the quotient by b_0 with a Markstein
    // correction, within 1.5 ulp of the reference's single division.
    const bool pow_norm = !opts.exact_division;
    std::string ap0x2; // 2 aP[0]
    const auto emit_pair_reads = [&](std::uint32_t k) {
        const auto rd = [&](std::size_t t) { return e.def(slabk(k, utname(t))); };
        return std::vector<std::string>{rd(pt.s0), rd(pt.s1), rd(pt.p0), rd(pt.p1)};
    };
    const auto emit_pair_compute = [&](std::uint32_t k, const std::vector<std::string> &rdv) {
        using emit_detail::ssa_emitter;
        const auto &es0 = rdv[0], &es1 = rdv[1], &ep0 = rdv[2], &ep1 = rdv[3];
        aS[k] = e.def(es0 + " - " + es1);
        const auto dP = e.def(ep0 + " - " + ep1);
        std::string sqS, sqy;
        if (k == 0u) {
            sqS = e.def(ssa_emitter::mul(aS[0], aS[0]));
            sqy = e.def(ssa_emitter::mul(dP, dP));
        } else {
            const auto acc4 = e.chain(hc4, aS[k], aS[0]);
            sqS = (k % 2u == 0u) ? e.def("__builtin_fma(2.0, " + acc4 + ", " + hmid + ")") : e.def(acc4 + " + " + acc4);
            // (A: 2 d_1[k] d_1[0] on top of the symmetric history sum; ap0x2 = 2 aP[0].)
            sqy = e.chain(hc2, dP, ap0x2);
        }
        // NOTE: role-dependent values are formed arithmetically with the lane constants fA / fB (1.0 on the lanes of
        // the role, 0.0 on the others) instead of selects (two v_cndmask per double): lane B reads the same slot twice
        // for the second difference, so that its dP is an exact zero.
        const auto mine = e.def("__builtin_fma(fA, " + sqy + ", " + sqS + ")");
        const auto oth = e.def("hy_swap1(" + mine + ")");
        const auto r2 = e.def(mine + " + " + oth);
        aP[k] = e.def("__builtin_fma(" + std::string((pow_norm && k >= 1u) ? "rb1n" : "fB") + ", " + r2 + ", " + dP + ")");
        std::string c1a;
        if (k == 0u) {
            // NOTE: the sum of squares is the same on both lanes: each of them evaluates the pow itself.
            const auto a0 = e.pow_eval(r2, pp.ex);
            aR[0] = pp.sc >= 0 ? e.def(ssa_emitter::mul(dtname(pt.csc), a0)) : a0;
            // (Zero on lane A: its quotient below is then an exact zero and sa_k = own + partner's.)
            rb1 = e.def("isB ? (1.0 / " + aP[0] + ") : 0.0");
            if (pow_norm) {
                os << "const double rb1n = " << rb1 << ";\n";
            }
            ap0x2 = e.def(aP[0] + " + " + aP[0]);
        } else {
            c1a = e.chain(hc1, aP[k], aR[0]);
            // NOTE: lane B keeps -(alpha + 1) j sa_j in aRp, so that its c2 chain is -(alpha + 1) S2 right away.
            std::string num;
            if (pow_norm) {
                const auto m = e.def(ssa_emitter::mul(fp_literal(pp.ex), c1a));
                const auto sab = hc2.empty() ? m
                                             : e.def("__builtin_fma(" + hc2 + ", " + fp_literal(1. / static_cast<double>(k)) + ", " + m + ")");
                // (Masked to the B lanes: the value doubles as the operand of aRp below.)
                const auto t = e.def(ssa_emitter::mul("fB", sab));
                aR[k] = e.def("hy_dpp<0xF5>(" + t + ")");
                if (k + 2u <= order) {
                    aRp[k] = e.def("__builtin_fma(" + fp_literal(-(pp.ex + 1.) * static_cast<double>(k)) + ", " + t + ", " + dP + ")");
                }
            } else {
            if (hc2.empty()) {
                num = e.def(ssa_emitter::mul(fp_literal(pp.ex * static_cast<double>(k)), c1a));
            } else {
                num = e.def(fp_literal(pp.ex * static_cast<double>(k)) + " * " + c1a + " + " + hc2);
            }
            // Division by k * b_0 (src/math/pow.cpp:546-549) without a division sequence on the critical path:
            // n = num * RN(1 / k), q0 = n * r with r = RN(1 / b_0); residual rem = n - b_0 * q0 (exact, FMA);
            // q = q0 + rem * r (Markstein: the correctly-rounded n / b_0 unless r is off by more than an ulp in a
            // halfway case; n itself carries the rounding of the scaling by 1 / k, so q is within 1.5 ulp of the
            // quotient num / (k b_0) the reference rounds once).
            const auto nk = (k == 1u) ? num : e.def(ssa_emitter::mul(num, fp_literal(1. / static_cast<double>(k))));
            const auto q0 = e.def(ssa_emitter::mul(nk, rb1));
            const auto rem = e.def("__builtin_fma(-" + aP[0] + ", " + q0 + ", " + nk + ")");
            const auto sab = e.def("__builtin_fma(" + rem + ", " + rb1 + ", " + q0 + ")");
            const auto sao = e.def("hy_swap1(" + sab + ")");
            aR[k] = e.def(sab + " + " + sao);
            }
        }
        if (!pow_norm && k >= 1u && k + 2u <= order) {
            const auto t = e.def(ssa_emitter::mul("fB", aR[k]));
            aRp[k] = e.def("__builtin_fma(" + fp_literal(-(pp.ex + 1.) * static_cast<double>(k)) + ", " + t + ", " + dP + ")");
        }
        std::string prS, prP;
        if (k == 0u) {
            prS = e.def(ssa_emitter::mul(aS[0], aR[0]));
            prP = e.def(ssa_emitter::mul(aP[0], aR[0]));
        } else {
            prS = e.chain(e.chain(hc3, aS[k], aR[0]), aS[0], aR[k]);
            prP = e.chain(c1a, aP[0], aR[k]);
        }
        os << slabk(k, utname(pt.os)) << " = " << prS << ";\n";
        os << slabk(k, utname(pt.op)) << " = " << prP << ";\n";
        if (has_rx && !fuse_rx) {
            const auto rS = e.def(ssa_emitter::mul(dtname(pt.crs), prS));
            const auto rP = e.def(ssa_emitter::mul(dtname(pt.crp), prP));
            os << slabk(k, utname(pt.rs)) << " = " << rS << ";\n";
            os << slabk(k, utname(pt.rp)) << " = " << rP << ";\n";
        }
        // History parts of order K = k + 1 (indices 1 .. k), the four chains interleaved term by term.
        hc1.clear();
        hc2.clear();
        hc3.clear();
        hc4.clear();
        hmid.clear();
        const auto K = k + 1u;
        if (K < order && K >= 2u) {
            const auto jmax = (K % 2u == 1u) ? (K - 1u) / 2u : (K - 2u) / 2u;
            for (std::uint32_t j = 1; j < K; ++j) {
                hc1 = e.chain(hc1, aP[K - j], aR[j]);
                hc2 = e.chain(hc2, aP[K - j], aRp[j]);
                hc3 = e.chain(hc3, aS[K - j], aR[j]);
                if (j <= jmax) {
                    hc4 = e.chain(hc4, aS[K - j], aS[j]);
                }
            }
            if (K % 2u == 0u) {
                hmid = e.def(ssa_emitter::mul(aS[K / 2u], aS[K / 2u]));
            }
        }
    };

    const auto emit_pair_order = [&](std::uint32_t k) { emit_pair_compute(k, emit_pair_reads(k)); };

    // ---- One-lane pair program ("v5") ----
    // Histories of a lane (SSA names by order): sD[i] = d_i (i = 0, 1, 2), sB = b_k / b_0 (k >= 1; b = sum of squares),
    // sA = sa = (scaled) pow. Chains of order k, history part = indices 1 .. k-1:
    //   q_i = sum_{j <= jmax} d_i[k-j] d_i[j]                      (half of the symmetric sum of d_i^2, see below)
    //   T   = sum_j sB[k-j] sA[j]                                  (S1 of the pow recurrence)
    //   U   = sum of the suffix sums of T's terms = sum_j j sB[k-j] sA[j]   (S2; terms taken in the order j = k-1 .. 1)
    //   c_i = sum_j d_i[k-j] sA[j]                                 (d_i * sa)
    // b_k = 2 (q_0 + q_1 + q_2) (+ the middle squares at even orders): the factor 2 is exact, so the HALF sum
    // bh = (q_0 + 0.5 mid_0) + ... is formed instead and the doubling is folded into the normalisation constant
    // rb2 = 2 RN(1 / b_0). a_k = alpha (T + sB[k] a_0) - ((alpha + 1) / k) U (src/math/pow.cpp:517-549 divided by k b_0).
    std::vector<std::string> sD[3], sB(order + 1u), sA(order + 1u);
    for (auto &v : sD) {
        v.resize(order + 1u);
    }
    std::string hq[3], hm[3], hcx[3], hT, hU, pow_pre;
    // (Tried in round 5 and removed: the stores of a round spread over the convolution chains which follow it instead of a
    // burst at the end of the dependent section - the eight wavefronts of a CU queue on one LDS store path -: -1.6 %,
    // profiles/r05_ab_spread_stores.log; the early chain terms of order k + 1 interleaved with the dependent operations of
    // round k: -1 %, profiles/r05_ab_interleaved_early_terms.log.)
    const auto emit_store = [&](const std::string &stmt) { os << stmt; };
    // Sensitivity experiment (profiles/experiments/sensitivity.py): HEYOKA_AMD_V5_PAD = "chain:dep:st:ld:salu" adds that many
    // dummy instructions of each kind to every order - independent FMAs in the chain section, dependent FMAs, LDS stores
    // and LDS reads in the dependent section, scalar no-ops - without touching the results: the slope of the step time
    // against each count says which resource the kernel is bound by.
    // (A sixth field: that many per-lane doubles kept live through the step, each used by one dependent FMA per order - what
    // per-lane coefficients of the acceleration sums would cost in registers; "norx" among HEYOKA_AMD_V5_OPTS drops the
    // reaction products and their stores - WRONG results, timing only: what fusing them into the sums could gain.)
    unsigned pad_chain = 0, pad_dep = 0, pad_st = 0, pad_ld = 0, pad_salu = 0, pad_regs = 0;
    if (!opts.dev.v5_pad.empty()) {
        std::sscanf(opts.dev.v5_pad.c_str(), "%u:%u:%u:%u:%u:%u", &pad_chain, &pad_dep, &pad_st, &pad_ld, &pad_salu, &pad_regs);
    }
    const bool any_pad = (pad_chain | pad_dep | pad_st | pad_ld | pad_salu | pad_regs) != 0u;
    const bool exp_norx = v5_flag("norx");
    // One accumulator for the half sum of squares bh_k = sum_i (sum_j d_i[k-j] d_i[j] + 1/2 d_i[k/2]^2): the three chains
    // of the coordinates run into each other - two additions per order and two multiply-adds per even order less, two
    // accumulators less. (The reference adds the three squares pairwise, src/detail/sum_sq.cpp:120-245: same terms, other
    // rounding order - inside the stated tolerances like the suffix sums of the pow recurrence.)
    const bool merged_sq = !v5_flag("nomsq");
    //   nosc      the selector's logarithm / exponential with literal polynomial constants (hy_sel_log(), exp()).
    const bool sel_scalar = one_lane && !v5_flag("nosc");
    // Issue priority (s_setprio): raised between the LDS exchange and the end of the finishing operations of a round - the
    // dependent chain which decides how soon the next exchange can start - and lowered for the convolution chains, so
    // that the wavefront which is in its critical section wins the VALU over the one streaming FMAs.
    // Measured (outer Solar System, 1 048 576 systems, A/B harness): 7.02e8 -> 7.15e8 system-steps/s; on by default,
    // HEYOKA_AMD_V5_PRIO=0 switches it off, =2 also keeps the serial tail of the step at the high priority.
    const int prio_mode = opts.dev.v5_prio;
    const bool prio_switch = prio_mode != 0;
    const auto emit_single_reads = [&](std::uint32_t k) {
        std::vector<std::string> r(6);
        if (vexch) {
            // The velocity coefficients of order k - 1 of the two bodies (row k - 1 of the jets; order 0: their current
            // positions, behind the rows).
            const auto row = k == 0u ? jet_rows_doubles : static_cast<std::uint64_t>(k - 1u) * spw * n_colp;
            for (std::uint32_t i = 0; i < 3u; ++i) {
                for (std::uint32_t sd = 0; sd < 2u; ++sd) {
                    r[2u * i + sd] = e.def("jetq[" + utname(st1.s[i][sd]) + " + " + std::to_string(row) + "u]");
                }
            }
            return r;
        }
        if (wide_rd) {
            // (x, y) of the two bodies with one ds_read_b128 each, then the two z.
            for (std::uint32_t sd = 0; sd < 2u; ++sd) {
                const auto w = wide_read(utname(st1.s[0][sd]));
                r[0u + sd] = w + ".x";
                r[2u + sd] = w + ".y";
            }
            for (std::uint32_t sd = 0; sd < 2u; ++sd) {
                r[4u + sd] = e.def(slabk(k, utname(st1.s[2][sd])));
            }
            return r;
        }
        for (std::uint32_t i = 0; i < 3u; ++i) {
            r[2u * i] = e.def(slabk(k, utname(st1.s[i][0])));
            r[2u * i + 1u] = e.def(slabk(k, utname(st1.s[i][1])));
        }
        return r;
    };
    const auto emit_single_compute = [&](std::uint32_t k, const std::vector<std::string> &rdv) {
        using emit_detail::ssa_emitter;
        for (std::uint32_t i = 0; i < 3u; ++i) {
            sD[i][k] = e.def(rdv[2u * i] + " - " + rdv[2u * i + 1u]);
            if (vexch && k >= 2u) {
                // (d^[k] = (v_a^[k-1] - v_b^[k-1]) RN(1 / k).)
                sD[i][k] = e.def(ssa_emitter::mul(sD[i][k], fp_literal(1. / static_cast<double>(k))));
            }
        }
        std::string pr[3];
        if (k == 0u) {
            // (Products rounded one by one, summed pairwise like the reference's sum_sq: src/detail/sum_sq.cpp:120-245.)
            std::string sq[3];
            for (std::uint32_t i = 0; i < 3u; ++i) {
                sq[i] = e.def(ssa_emitter::mul(sD[i][0], sD[i][0]));
            }
            const auto s01 = e.def(sq[0] + " + " + sq[1]);
            const auto r2 = e.def(s01 + " + " + sq[2]);
            sB[0] = r2;
            const auto a0 = e.pow_eval(r2, pp.ex);
            sA[0] = pp.sc >= 0 ? e.def(ssa_emitter::mul(dtname(st1.csc), a0)) : a0;
            const auto rb = e.def("1.0 / " + r2);
            os << "const double rb2 = " << rb << " + " << rb << ";\n";
            os << "const double arbA = " << fp_literal(pp.ex) << " * (rb2 * " << sA[0] << ");\n";
            for (std::uint32_t i = 0; i < 3u; ++i) {
                pr[i] = e.def(ssa_emitter::mul(sD[i][0], sA[0]));
            }
        } else {
            // The dependent chain from the exchange to the stores decides how soon the next round can start, so
            // everything which does not need an order-k input was folded into the history accumulators at the end of the
            // previous round (the middle squares into hq, alpha T - ((alpha + 1) / k) U into pow_pre): what is left is
            // sub -> fma -> add -> add -> fma (sa_k) -> fma (products) -> mul (reactions).
            std::string bh;
            if (merged_sq) {
                bh = hq[0];
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    bh = e.chain(bh, sD[i][k], sD[i][0]);
                }
            } else {
                std::string q[3];
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    q[i] = e.chain(hq[i], sD[i][k], sD[i][0]);
                }
                const auto q01 = e.def(q[0] + " + " + q[1]);
                bh = e.def(q01 + " + " + q[2]);
            }
            // sa_k = alpha (T + (b_k / b_0) sa_0) - ((alpha + 1) / k) U with b_k / b_0 = rb2 bh: alpha rb2 sa_0 is a constant
            // of the step (arbA).
            sA[k] = pow_pre.empty() ? e.def(ssa_emitter::mul(bh, "arbA")) : e.def("__builtin_fma(" + bh + ", arbA, " + pow_pre + ")");
            sB[k] = e.def(ssa_emitter::mul("rb2", bh));
            for (std::uint32_t i = 0; i < 3u; ++i) {
                pr[i] = e.chain(e.chain(hcx[i], sD[i][k], sA[0]), sD[i][0], sA[k]);
            }
        }
        for (std::uint32_t i = 0; i < 3u; ++i) {
            emit_store(slabk(k, utname(st1.o[i])) + " = " + pr[i] + ";\n");
        }
        for (std::uint32_t i = 0; pp.rx[0] >= 0 && !fuse_rx && i < 3u && !exp_norx; ++i) {
            // (The reaction on the second body of the pair: c * (d_i * sa), src/model/nbody.cpp:113-130.)
            const auto rxv = e.def(ssa_emitter::mul("crs_r", pr[i]));
            emit_store(slabk(k, utname(st1.r[i])) + " = " + rxv + ";\n");
        }
        if (any_pad && k >= 1u) {
            for (unsigned i = 0; i < pad_dep; ++i) {
                os << "asm volatile(\"v_fma_f64 %0, %0, %0, %0\" : \"+v\"(hy_pad0));\n";
            }
            for (unsigned i = 0; i < pad_regs; ++i) {
                os << "asm volatile(\"v_fma_f64 %0, %1, %0, %0\" : \"+v\"(hy_pad0) : \"v\"(hy_rp" << i << "));\n";
            }
            // (Written-out LDS stores of the first product to its own slot once more: same value, same address. The
            // compiler's lgkmcnt bookkeeping stays conservative: LDS operations complete in order.)
            for (unsigned i = 0; i < pad_st % 100u; ++i) {
                os << "asm volatile(\"ds_write_b64 %0, %1\" ::\"v\"((unsigned)(unsigned long long)&" << slabk(k, utname(st1.o[0]))
                   << "), \"v\"(" << pr[0] << ") : \"memory\");\n";
            }
            // (pad_st >= 100: 16-byte stores - the first product and its neighbour in the array written back as a pair.)
            for (unsigned i = 0; i < pad_st / 100u; ++i) {
                os << "{\nhy_d2 hy_pv;\nhy_pv.x = " << pr[0] << ";\nhy_pv.y = " << pr[0] << ";\n"
                   << "asm volatile(\"ds_write_b128 %0, %1\" ::\"v\"((unsigned)(unsigned long long)(slab + (" << dummy_base
                   << "u & ~1u))), \"v\"(hy_pv) : \"memory\");\n}\n";
            }
            // (LDS reads of the velocity coefficients of the previous order once more, summed into a dummy: the reads of
            // this round cannot be shared with them - the wave barrier between the rounds is a memory clobber.)
            for (unsigned i = 0; vexch && k >= 2u && i < pad_ld && i < 6u; ++i) {
                os << "hy_pad4 = hy_pad4 + jetq[" << utname(st1.s[i % 3u][i / 3u]) << " + "
                   << static_cast<std::uint64_t>(k - 2u) * spw * n_colp << "u];\n";
            }
            for (unsigned i = 0; i < pad_salu; ++i) {
                os << "asm volatile(\"s_nop 0\");\n";
            }
        }
        if (prio_switch) {
            // (End of the latency-critical part of the round: the chains below are bulk work.)
            os << "__builtin_amdgcn_s_setprio(0);\n";
        }
        if (any_pad && k >= 1u) {
            for (unsigned i = 0; i < pad_chain; ++i) {
                os << "asm volatile(\"v_fma_f64 %0, %0, %0, %0\" : \"+v\"(hy_pad" << (1u + i % 4u) << "));\n";
            }
        }
    };
    const auto emit_single_history = [&](std::uint32_t k) {
        using emit_detail::ssa_emitter;
        // History parts of order K = k + 1 (terms without an order-K operand) and the T / U chains of the pow recurrence,
        // whose first term is the newest one.
        for (std::uint32_t i = 0; i < 3u; ++i) {
            hq[i].clear();
            hm[i].clear();
            hcx[i].clear();
        }
        hT.clear();
        hU.clear();
        const auto K = k + 1u;
        if (K < order && K >= 2u) {
            const auto jmax = (K % 2u == 1u) ? (K - 1u) / 2u : (K - 2u) / 2u;
            for (std::uint32_t j = 1; j < K; ++j) {
                // (T / U take their terms in the order j = K-1 .. 1: the first term enters U K-1 times, the last one once.)
                const auto jd = K - j;
                hT = e.chain(hT, sB[K - jd], sA[jd]);
                hU = hU.empty() ? hT : e.def(hU + " + " + hT);
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    hcx[i] = e.chain(hcx[i], sD[i][K - j], sA[j]);
                    if (j <= jmax) {
                        auto &acc = hq[merged_sq ? 0u : i];
                        acc = e.chain(acc, sD[i][K - j], sD[i][j]);
                    }
                }
            }
            if (K % 2u == 0u) {
                for (std::uint32_t i = 0; i < 3u; ++i) {
                    if (merged_sq) {
                        // (One running sum of the three middle squares.)
                        hm[0] = e.chain(i == 0u ? std::string{} : hm[0], sD[i][K / 2u], sD[i][K / 2u]);
                    } else {
                        hm[i] = e.def(ssa_emitter::mul(sD[i][K / 2u], sD[i][K / 2u]));
                    }
                }
            }
            // Off the critical path of round K: the middle squares join the half sums, the two sums of the pow
            // recurrence are combined.
            if (K % 2u == 0u) {
                for (std::uint32_t i = 0; i < (merged_sq ? 1u : 3u); ++i) {
                    hq[i] = hq[i].empty() ? e.def(ssa_emitter::mul("0.5", hm[i]))
                                          : e.def("__builtin_fma(0.5, " + hm[i] + ", " + hq[i] + ")");
                }
            }
            const auto t1 = e.def(ssa_emitter::mul(fp_literal(-(pp.ex + 1.) / static_cast<double>(K)), hU));
            pow_pre = e.def("__builtin_fma(" + fp_literal(pp.ex) + ", " + hT + ", " + t1 + ")");
        } else {
            pow_pre.clear();
        }
    };
    // External inputs which are constant u variables in EVERY cluster (isomorphic clusters may pair a constant with a
    // variable: the heliocentric alias x_i - 0 and the pair difference x_j - x_i of model::np1body).
    std::vector<char> ext_const(n_ext, 1);
    for (std::uint32_t x = 0; x < n_ext; ++x) {
        for (std::size_t c = 0; c < nc; ++c) {
            ext_const[x] = (ext_const[x] != 0 && cu[pl.ext_u[c][x]] != 0) ? 1 : 0;
        }
    }
    const auto emit_cluster = [&](std::uint32_t k) {
        if (pair_split) {
            emit_pair_order(k);
            return;
        }
        for (std::uint32_t part = 1; part < n_parts; ++part) {
            e.emit_partials(t0_ids, k, part, n_parts);
        }
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            if (ext_const[x] != 0 && k > 0u) {
                // A constant input of every cluster (e.g. -par[i]): read once, at order 0.
                e.val(pl.ext_u[0][x], k) = "0.0";
                continue;
            }
            e.val(pl.ext_u[0][x], k) = e.def(slabk(k, utname(ext_tbl[x])));
        }
        if (overlap) {
            sched_fence();
            e.emit_partials_sel(t0_ids, k + 1u, psel::early_b);
            if (fence2) {
                sched_fence();
            }
        }
        for (const auto u : t0) {
            e.node_finish(u - n_eq, k);
        }
        for (std::uint32_t x = 0; x < n_out; ++x) {
            os << slabk(k, utname(out_tbl[x])) << " = " << e.val(t0[pl.out_pos[x]], k) << ";\n";
        }
    };

    // ===================== step body =====================
    os << "double m0 = 0.0, mo = 0.0, mom1 = 0.0;\n";
    if (frx_regs) {
        for (const auto &rg : rounds) {
            for (const auto &gr : rg) {
                for (const auto t : gr.coef_tbl) {
                    os << "const double frc" << t << " = " << dtname(t) << ";\n";
                }
            }
        }
    }
    if (one_lane && !opts.dev.v5_pad.empty()) {
        os << "double hy_pad0 = 1.0, hy_pad1 = 1.0, hy_pad2 = 1.0, hy_pad3 = 1.0, hy_pad4 = 1.0;\n";
        unsigned n_rp = 0;
        std::sscanf(opts.dev.v5_pad.c_str(), "%*u:%*u:%*u:%*u:%*u:%u", &n_rp);
        for (unsigned i = 0; i < n_rp; ++i) {
            os << "double hy_rp" << i << " = lds_fac[(threadIdx.x + " << i << "u) % 40u];\n";
        }
    }
    for (auto &rg : rounds) {
        for (auto &gr : rg) {
            for (auto &ow : gr.owners) {
                os << "const double xs" << ow.col << " = " << row0_r(ow) << ";\n";
                publish_sv(ow, 0, "xs" + std::to_string(ow.col));
            }
        }
    }
    // (One-lane pair kernel, single-buffered slab: the order-1 coefficients go to the slab once the order-0 ones have
    // been read, i.e. after the reads of round 0.)
    std::vector<std::tuple<owner_slot *, std::uint32_t, std::string>> deferred_pub;
    if (merged) {
        // Orders 1 .. a of the a-th variable of a chain follow from the state alone.
        for (auto &rg : rounds) {
            for (auto &gr : rg) {
                for (std::size_t a = 1; a < gr.owners.size(); ++a) {
                    for (std::uint32_t j = 1; j <= a && j <= order; ++j) {
                        const auto x = e.div_const(gr.owners[a - 1u].xname[j - 1u], j);
                        if (one_lane) {
                            deferred_pub.emplace_back(&gr.owners[a], j, x);
                        } else {
                            publish_sv(gr.owners[a], j, x);
                        }
                    }
                }
            }
        }
    }
    sync();
    for (std::uint32_t k = 0; merged && k <= order; ++k) {
        std::vector<std::string> prd;
        if (k < order) {
            prd = one_lane ? emit_single_reads(k) : emit_pair_reads(k);
        }
        if (k == 0u) {
            for (auto &[ow, j, x] : deferred_pub) {
                publish_sv(*ow, j, x);
            }
        }
        std::vector<std::tuple<std::size_t, std::uint32_t, std::vector<std::string>>> pend;
        if (k >= 1u) {
            for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                for (std::uint32_t r = 0; r < rounds[g].size(); ++r) {
                    pend.emplace_back(g, r, emit_glue_reads(g, r, k - 1u));
                }
            }
        }
        if (one_lane) {
            // Round k of the one-lane kernel: LDS reads | early chains of order k + 1 (independent of the reads) | glue of
            // order k - 1 (consumes its ten operands right away: 20 registers which would otherwise stay live across
            // the finishing operations of the pairs) | finishing of order k, stores, late chains.
            if (prio_switch) {
                os << "__builtin_amdgcn_s_setprio(3);\n";
            }
            sched_fence();
            if (k < order) {
                sched_fence();
            }
            for (const auto &[g, r, names] : pend) {
                emit_glue_compute(g, r, k - 1u, names);
            }
            if (k < order) {
                emit_single_compute(k, prd);
            }
            if (k < order) {
                emit_single_history(k);
            }
            if (prio_switch && k == order && prio_mode != 2) {
                os << "__builtin_amdgcn_s_setprio(0);\n";
            }
            sync();
            continue;
        }
        if (k < order) {
            emit_pair_compute(k, prd);
        }
        for (const auto &[g, r, names] : pend) {
            emit_glue_compute(g, r, k - 1u, names);
        }
        sync();
    }
    for (std::uint32_t k = 0; !merged && k < order; ++k) {
        for (std::uint32_t lev = 1; lev <= pl.max_level; ++lev) {
            if (lev == pl.cluster_level) {
                emit_cluster(k);
            }
            const bool last = (lev == pl.max_level);
            if (overlap && last) {
                // Reads of all the rounds of the level first, then the independent chunk, then the computations.
                std::vector<std::tuple<std::size_t, std::uint32_t, std::vector<std::string>>> pend;
                for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                    if (pl.groups[g].level == lev) {
                        for (std::uint32_t r = 0; r < rounds[g].size(); ++r) {
                            pend.emplace_back(g, r, emit_glue_reads(g, r, k));
                        }
                    }
                }
                if (!pair_split) {
                    sched_fence();
                    e.emit_partials_sel(t0_ids, k + 1u, psel::late);
                    e.emit_partials_sel(t0_ids, k + 2u, psel::early_a);
                    if (fence2) {
                        sched_fence();
                    }
                }
                for (const auto &[g, r, names] : pend) {
                    emit_glue_compute(g, r, k, names);
                }
            } else {
                for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                    if (pl.groups[g].level == lev) {
                        for (std::uint32_t r = 0; r < rounds[g].size(); ++r) {
                            emit_glue_round(g, r, k);
                        }
                    }
                }
                if (!pair_split && !overlap && last && k + 1u < order) {
                    // History part of the next order's convolutions: overlaps the exchange latency.
                    e.emit_partials(t0_ids, k + 1u, 0, n_parts);
                }
            }
            sync();
        }
    }
    if (one_lane && !opts.dev.v5_pad.empty()) {
        os << "asm volatile(\"\" ::\"v\"(hy_pad4));\n";
    }
    const auto body = os.str();
    os.str("");
    os.clear();

    // NOTE: a local search over slot permutations minimising the LDS bank conflicts of the gather-type reads
    // was tried and removed: it lowered SQ_LDS_BANK_CONFLICT by 9 % with no change in kernel time. A
    // microbenchmark on gfx950 shows why: a single wavefront issues one ds_read_b64 per 6.6 clk whatever the
    // pattern (consecutive, 2-way conflicting, strided groups), only a 64-way same-bank pattern is slower
    // (34.6 clk); the cost of the LDS traffic here is the issue slots of its ~620 instructions per step.

    // ===================== module text =====================
    std::ostringstream src;
    src << "#define SPW " << spw << "u\n#define HY_WPB " << wpb << "u\n";
    src << "#define HY_M4 " << (m4 ? 1 : 0) << "\n";
    // (The stepper with events is a specialisation of its own and always runs in mode 4: a compile-time constant there - the
    // bookkeeping of the propagation mode, ~900 instructions per group of systems around a single step, is not generated.)
    src << "#define HY_MODE " << (m4 ? "4" : "a.mode") << "\n";
    // (Propagation from grid point to grid point: see hy_kargs::tc_thr.)
    src << "#define HY_GRID_STOP " << ((one_lane && jet_lds && !m4) ? "((HY_MODE == 1) & ((a.pad & 4) != 0) & hy_reach)" : "false") << "\n";
    src << "#define HY_GRID_FLAGS " << ((one_lane && jet_lds && !m4) ? "((HY_MODE == 1) & ((a.pad & 4) != 0))" : "false") << "\n";
    // (The stepper with events always uses the static schedule: its cooperative store has workgroup barriers.)
    src << "#define HY_NO_STATIC 0\n";
    src << prelude;
    emit_detail::emit_dout(src, p, opts);
    src << emit_detail::wsync_macro;
    src << R"HIP(
// Maximum of the norm accumulators. They start from 0.0 and (a < b) ? b : a never selects a NaN b: the first operand
// is never a NaN, and for such operands the IEEE maximum (one instruction: it returns the non-NaN operand) is the same
// function as the comparison + select of hy_max(). (Written as the instruction itself: fmax() would first canonicalise
// every operand which comes out of a lane exchange, one more v_max_f64 each.)
__device__ __forceinline__ double hy_nmax(double a, double b)
{
#if defined(HY_NO_NMAX)
    return hy_max(a, b);
#else
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
// A product which is never contracted into an FMA with its consumer (the value must be the ROUNDED product).
__device__ __forceinline__ double hy_mul_nc(double x, double y)
{
    double t = x * y;
    asm("" : "+v"(t));
    return t;
}
template <int CTRL>
__device__ __forceinline__ double hy_dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
)HIP";
    // (hy_sel_log(): the logarithm of the step-size selector, in the common prelude - hip_emit.cpp.)
    if (wide_rd) {
        // (The native vector type: a load of HIP's double2 - a struct - is taken apart into two 8-byte loads by SROA.)
        src << "#if defined(HY_HOST_EMU)\nstruct hy_d2 {\n    double x, y;\n};\n#else\n"
               "typedef double hy_d2 __attribute__((ext_vector_type(2)));\n#endif\n";
    }
    if (one_lane) {
        // The logarithm and the exponential of the selector with their polynomial constants in SCALAR registers. A Horner
        // step p * w + C with a literal C compiles to v_mov_b32 x 2 (the literal into the destination pair) + v_fmac_f64:
        // three VALU instructions where v_fma_f64 with C in an SGPR pair is one - VOP3 takes no 64-bit literal, and the
        // instruction selector prefers the two-address form even when the constant already sits in scalar registers, hence
        // the instruction is written out (hy_fma_sc). The constants are materialised by s_mov_b32 pairs next to their use
        // (MachineLICM is off for this module), which issue on the scalar unit next to the other wavefront's arithmetic: 39 VALU instructions per step less (20 in the logarithm, 19 against the device library's
        // exp(), whose minimax polynomial has the same shape). hy_sel_exp_s(): k = rint(x / ln 2), r = x - k ln 2 in two
        // pieces, Taylor polynomial of degree 13 on |r| <= 0.347 (truncation 4e-18), ldexp; +inf -> +inf, -inf -> 0, nan -> nan.
        src << R"HIP(
#if defined(HY_HOST_EMU)
#define hy_fma_sc(a, b, c) __builtin_fma((a), (b), (c))
#else
// a * b + c with the (uniform) addend c in a scalar register pair: ONE v_fma_f64.
__device__ __forceinline__ double hy_fma_sc(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#endif
__device__ __forceinline__ double hy_sel_log_s(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0x1.6a09e667f3bcdp-1;
    m = m * (lo ? 2.0 : 1.0);
    e -= lo ? 1 : 0;
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    double z = num * r;
    z = __builtin_fma(__builtin_fma(-den, z, num), r, z);
    const double w = z * z;
    double p = __builtin_fma(w, 0x1.8618618618618p-4, 0x1.af286bca1af28p-4);
    p = hy_fma_sc(p, w, 0x1.e1e1e1e1e1e1ep-4);
    p = hy_fma_sc(p, w, 0x1.1111111111111p-3);
    p = hy_fma_sc(p, w, 0x1.3b13b13b13b14p-3);
    p = hy_fma_sc(p, w, 0x1.745d1745d1746p-3);
    p = hy_fma_sc(p, w, 0x1.c71c71c71c71cp-3);
    p = hy_fma_sc(p, w, 0x1.2492492492492p-2);
    p = hy_fma_sc(p, w, 0x1.999999999999ap-2);
    p = hy_fma_sc(p, w, 0x1.5555555555555p-1);
    const double ed = (double)e;
    double res = __builtin_fma(ed, 0x1.abc9e3b39803fp-56, (z * w) * p);
    res = __builtin_fma(2.0, z, res);
    res = __builtin_fma(ed, 0x1.62e42fefa39efp-1, res);
    res = (x == 0.0) ? -__builtin_inf() : res;
    res = (x == __builtin_inf()) ? x : res;
    return res;
}
__device__ __forceinline__ double hy_sel_exp_s(double x)
{
    const double kf = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(kf, -0x1.62e42fefa39efp-1, x);
    r = __builtin_fma(kf, -0x1.abc9e3b39803fp-56, r);
    double p = __builtin_fma(r, 0x1.6124613a86d09p-33, 0x1.1eed8eff8d898p-29);
    p = hy_fma_sc(p, r, 0x1.ae64567f544e4p-26);
    p = hy_fma_sc(p, r, 0x1.27e4fb7789f5cp-22);
    p = hy_fma_sc(p, r, 0x1.71de3a556c734p-19);
    p = hy_fma_sc(p, r, 0x1.a01a01a01a01ap-16);
    p = hy_fma_sc(p, r, 0x1.a01a01a01a01ap-13);
    p = hy_fma_sc(p, r, 0x1.6c16c16c16c17p-10);
    p = hy_fma_sc(p, r, 0x1.1111111111111p-7);
    p = hy_fma_sc(p, r, 0x1.5555555555555p-5);
    p = hy_fma_sc(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    double res = ldexp(p, (int)kf);
    res = (x > 710.0) ? __builtin_inf() : res;
    res = (x < -746.0) ? 0.0 : res;
    return res;
}
)HIP";
    }
    if (pair_split) {
        // Exchange between the two lanes of a pair: DPP quad_perm [1,0,3,2] on the two halves of the double.
        src << R"HIP(
__device__ __forceinline__ double hy_swap1(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
)HIP";
    }
    src << "__constant__ unsigned short hy_utbl[" << std::max<std::size_t>(utbl.size(), 1u) * L << "] = {";
    for (const auto &v : utbl) {
        for (const auto x : v) {
            if (x > 65535u) {
                why_not = "slot / variable index overflow in the lane tables";
                return ret;
            }
            src << x << ",";
        }
    }
    src << "};\n";
    {
        // Jet column of every state variable.
        std::vector<std::uint32_t> col_of(n_eq, 0);
        for (const auto &rg : rounds) {
            for (const auto &gr : rg) {
                for (const auto &ow : gr.owners) {
                    const auto &vv = utbl[ow.var_tbl];
                    for (std::uint32_t l2 = 0; l2 < ow.n_valid; ++l2) {
                        col_of[vv[l2]] = ow.cbase + l2;
                    }
                }
            }
        }
        if (m4) {
            // Source table of the cooperative store of the Taylor coefficients (mode 4): per stored row its index in a.tc
            // (variable * (order + 1) + k), the offset of its value for the first system of a wavefront inside the jets of
            // the wavefront, and the stride between the systems of the wavefront. Compact set (emitted_module::compact_tc):
            // a state variable defined by another state variable (x' = v) leaves with its order-0 row only.
            struct tc_entry {
                std::uint64_t row, off, stride;
            };
            std::vector<tc_entry> tc_src;
            if (one_lane) {
                // (Velocity-type jets [row][owner slot][system][lane]; current values of the derived variables behind them.)
                for (const auto &rg : rounds) {
                    for (const auto &gr : rg) {
                        for (const auto &ow : gr.owners) {
                            const auto &vv = utbl[ow.var_tbl];
                            for (std::uint32_t l2 = 0; l2 < ow.n_valid; ++l2) {
                                const std::uint64_t var = vv[l2];
                                if (ow.derived) {
                                    tc_src.push_back({var * (order + 1u), jet_rows_doubles + jet_off(ow) + l2, jet_sys(ow)});
                                } else {
                                    for (std::uint32_t k = 0; k <= order; ++k) {
                                        tc_src.push_back({var * (order + 1u) + k,
                                                          static_cast<std::uint64_t>(k) * spw * n_colp + jet_off(ow) + l2, jet_sys(ow)});
                                    }
                                }
                            }
                        }
                    }
                }
            } else {
                for (std::uint32_t var = 0; var < n_eq; ++var) {
                    const auto &sd = p.sv_defs[var];
                    const bool derived = compact_tc && sd.type == operand::kind::uvar && sd.idx < n_eq;
                    for (std::uint32_t k = 0; k <= (derived ? 0u : order); ++k) {
                        tc_src.push_back({static_cast<std::uint64_t>(var) * (order + 1u) + k,
                                          static_cast<std::uint64_t>(k) * spw * n_colp + col_of[var], n_colp});
                    }
                }
            }
            n_tc_rows = tc_src.size();
            src << "__constant__ unsigned long long hy_tc_src[" << std::max<std::size_t>(n_tc_rows, 1u) << "] = {";
            for (const auto &t : tc_src) {
                if (t.row >= (1ull << 20) || t.off >= (1ull << 20) || t.stride >= (1ull << 20)) {
                    why_not = "mode 4: index overflow in the source table of the Taylor coefficients";
                    return ret;
                }
                src << (t.row | (t.off << 20) | (t.stride << 40)) << "ull,";
            }
            src << "};\n";
        }
        src << "__constant__ unsigned short hy_col_of_var[" << n_eq << "] = {";
        for (const auto c : col_of) {
            src << c << ",";
        }
    }
    if (one_lane) {
        src << "};\n__constant__ double hy_rk[" << (order + 1u) << "] = {0.0,";
        for (std::uint32_t k = 1; k <= order; ++k) {
            src << fp_literal(1. / static_cast<double>(k)) << ",";
        }
    }
    src << "};\n__constant__ double hy_dtbl[" << std::max<std::size_t>(dtbl.size(), 1u) * L << "] = {";
    for (const auto &v : dtbl) {
        for (const auto x : v) {
            src << fp_literal(x) << ",";
        }
    }
    src << "};\n";

    src << "extern \"C\" __global__ void __launch_bounds__(" << bs << ") hy_taylor(const hy_kargs a)\n{\n";
    src << "__shared__ " << (wide_rd ? "__attribute__((aligned(16))) " : "") << "double lds_slab["
        << static_cast<std::uint64_t>(wpb) * spw * slab_stride << "];\n";
    src << "const unsigned lane = threadIdx.x & 63u;\nconst unsigned wib = threadIdx.x >> 6;\n";
    src << "const unsigned l = lane % " << L << "u;\nconst unsigned q = lane / " << L << "u;\n";
    src << "const u64 N = a.N;\n";
    if (pair_split) {
        src << "const bool isB = (lane & 1u) != 0u;\nconst double fB = isB ? 1.0 : 0.0, fA = isB ? 0.0 : 1.0;\n";
    }
    src << "double *const slab = lds_slab + (wib * " << spw << "u + q) * " << slab_stride << "u;\n";
    if (one_lane) {
        if (bk_in_slab) {
            src << "double *const bk = slab + " << (slab_stride - 16u) << "u;\n";
        } else {
            src << "__shared__ double lds_bk[" << wpb * spw * 16u << "];\ndouble *const bk = lds_bk + (wib * " << spw << "u + q) * 16u;\n";
        }
    }
    const bool vexch_decl = vexch;
    src << "const u64 gwave = (u64)blockIdx.x * " << wpb << "u + wib;\n";
    if (jet_lds) {
        src << "__shared__ double lds_jet[" << wpb * jet_doubles_per_wave << "];\n";
        src << "double *const jetw = lds_jet + wib * " << jet_doubles_per_wave << "u;\n";
        if (vexch_decl) {
            // (The rows of the system of this lane: [system][column].)
            src << "const double *const jetq = jetw + q * " << n_col << "u;\n";
        }
        if (m4) {
            // Source table of the cooperative store of the Taylor coefficients: row | jet column << 16.
            src << "__shared__ unsigned long long lds_tcsrc[" << std::max<std::size_t>(n_tc_rows, 1u) << "];\n";
            src << "for (unsigned i = threadIdx.x; i < " << n_tc_rows << "u; i += " << bs
                << "u) lds_tcsrc[i] = hy_tc_src[i];\n__syncthreads();\n";
        }
    } else {
        src << "double *const jetw = a.scratch + gwave * " << jet_doubles_per_wave << "ull;\n";
    }
    // The lane tables (state variable of every owner slot and lane) are read from LDS: in constant memory every lookup at
    // the pickup of a group and ahead of every store of its results is a vector load, and gfx9 counts loads and stores in
    // ONE in-order counter (vmcnt) - the address of each group of stores then waits for the acknowledgement of all the
    // stores before it (four owner slots: four round trips per group of systems; with the table in LDS: none).
    {
        const auto n_ut = std::max<std::size_t>(utbl.size(), 1u) * L;
        src << "__shared__ unsigned short lds_utbl[" << n_ut << "];\n";
        src << "for (unsigned i = threadIdx.x; i < " << n_ut << "u; i += " << bs << "u) lds_utbl[i] = hy_utbl[i];\n";
        src << "__syncthreads();\n#define hy_utbl lds_utbl\n";
    }
    for (std::size_t t = 0; t < utbl.size(); ++t) {
        if (utexpr[t] == "ut" + std::to_string(t)) {
            src << "const unsigned ut" << t << " = hy_utbl[" << t * L << "u + l];\n";
        }
    }
    if (one_lane) {
        src << "__shared__ double lds_dt[" << std::max<std::size_t>(dtbl.size(), 1u) * L << "];\n";
        src << "for (unsigned i = threadIdx.x; i < " << dtbl.size() * L << "u; i += " << bs << "u) lds_dt[i] = hy_dtbl[i];\n";
        src << "__syncthreads();\nconst double *const dtl = lds_dt + l;\n";
        if (pp.rx[0] >= 0 && !fuse_rx) {
            src << "const double crs_r = hy_dtbl[" << st1.crs * L << "u + l];\n";
        }
    }
    if (!pk_tbl.empty()) {
        // (Factors of the packed evaluation: row 0 = ones - the series of a variable with a jet column -, row 1 = RN(1 / k).)
        src << "__shared__ double lds_fac[" << 2u * (order + 1u) << "];\n";
        src << "for (unsigned i = threadIdx.x; i < " << 2u * (order + 1u) << "u; i += " << bs << "u) lds_fac[i] = (i <= " << order
            << "u) ? 1.0 : hy_rk[i - " << (order + 1u) << "u];\n__syncthreads();\n";
    }
    for (std::size_t t = 0; !one_lane && t < dtbl.size(); ++t) {
        src << "const double dt" << t << " = hy_dtbl[" << t * L << "u + l];\n";
    }
    for (std::uint32_t x = 0; !pairk && x < n_cst; ++x) {
        src << "const double ccst" << x << " = dt" << cst_tbl[x] << ";\n";
    }
    for (const auto &rg : rounds) {
        for (const auto &gr : rg) {
            for (const auto &ow : gr.owners) {
                src << "const bool ovalid" << ow.col << " = l < " << gr.n_valid << "u;\n";
                if (ow.derived) {
                    // (Current values of the derived variables: after the jet rows of the wavefront. One pointer for
                    // reading and writing: the idle lanes of a partially filled slot replicate the work of lane 0 bit by
                    // bit and store the same values to the same entry.)
                    src << "double *const x0c" << ow.col << " = jetw + " << jet_rows_doubles + jet_off(ow) << "u + q * " << jet_sys(ow)
                        << "u + (ovalid" << ow.col << " ? l : 0u);\n";
                    src << "const double *const x0r" << ow.col << " = x0c" << ow.col << ";\n";
                    continue;
                }
                if (one_lane) {
                    src << "double *const jc" << ow.col << " = jetw + " << jet_off(ow) << "u + q * " << jet_sys(ow) << "u + (ovalid"
                        << ow.col << " ? l : 0u);\n";
                    src << "const double *const jr" << ow.col << " = jc" << ow.col << ";\n";
                    continue;
                }
                src << "double *const jc" << ow.col << " = jetw + q * " << n_colp << "u + (ovalid" << ow.col << " ? "
                    << ow.cbase << "u + l : " << n_col << "u);\n";
                // The current state (order-0 row) is read from the column of the replicated variable by the idle lanes.
                src << "const double *const jr" << ow.col << " = jetw + q * " << n_colp << "u + " << ow.cbase
                    << "u + (ovalid" << ow.col << " ? l : 0u);\n";
            }
        }
    }
    // Lane slots of the final evaluation: slot h, lane l <-> jet column h * L + l (dummy column beyond the last one).
    for (std::uint32_t h = 0; !one_lane && h < n_hslots; ++h) {
        src << "double *const hc" << h << " = jetw + q * " << n_colp << "u + ((" << h * L << "u + l < " << n_col << "u) ? "
            << h * L << "u + l : " << n_col << "u);\n";
    }
    // Event equations inside the stepper (emit_options::ev_prog; one-lane-per-pair kernel): their jets from the jets of the
    // state variables in LDS - every lane of a system runs the same statements; the terms of a sum of isomorphic terms
    // (a squared distance, a radial velocity) are evaluated side by side, term c by lane c (ev_lane_hooks) -, the three
    // norms extended to them, then the ordinary selector, final evaluation and state update of this kernel. What is left
    // to the kernels behind the stepper: event detection on a.ev_tc, times / outcomes / records (hy_ev_post). The
    // statements are generated here (their per-lane offsets are declared ahead of the work loop) and pasted into the tail.
    bool ev_inline = false;
    // (lane -> (event, constant) of the close-encounter events which the lane of a pair contributes itself.)
    std::map<std::uint32_t, std::pair<std::uint32_t, double>> pe_lane_ev;
    std::map<std::uint32_t, double> pe_lane_sign; // (-1: the event equation is c - |r_i - r_j|^2)
    std::string ev_code;
    std::vector<std::vector<std::string>> ev_coeffs;
    const bool packed_tail_ev = L >= 4u;
    if (m4 && one_lane && jet_lds && packed_tail_ev && opts.ev_prog != nullptr && !opts.exact_division && slab_stride >= n_own * L
        && opts.dev.events_in_stepper) {
        struct sv_loc {
            bool derived = false;
            std::uint64_t off = 0;
            std::uint32_t stride = 0;
            std::uint32_t parent = 0;
        };
        std::vector<sv_loc> loc(n_eq);
        for (const auto &rg : rounds) {
            for (const auto &gr : rg) {
                for (const auto &ow : gr.owners) {
                    const auto &vv = utbl[ow.var_tbl];
                    for (std::uint32_t l2 = 0; l2 < ow.n_valid; ++l2) {
                        auto &lc = loc[vv[l2]];
                        lc.stride = jet_sys(ow);
                        if (ow.derived) {
                            lc.derived = true;
                            lc.off = jet_rows_doubles + jet_off(ow) + l2;
                            for (const auto &o2 : gr.owners) {
                                if (o2.col == ow.parent) {
                                    lc.parent = utbl[o2.var_tbl][l2];
                                }
                            }
                        } else {
                            lc.off = jet_off(ow) + l2;
                        }
                    }
                }
            }
        }
        const auto rowst = static_cast<std::uint64_t>(spw) * n_colp;
        const std::function<std::string(std::uint32_t, std::uint32_t)> sv = [&](std::uint32_t i, std::uint32_t k) -> std::string {
            const auto &lc = loc[i];
            if (!lc.derived) {
                return "jetw[" + std::to_string(k * rowst + lc.off) + "u + q * " + std::to_string(lc.stride) + "u]";
            }
            if (k == 0u) {
                return "jetw[" + std::to_string(lc.off) + "u + q * " + std::to_string(lc.stride) + "u]";
            }
            // x^[k] = v^[k-1] * RN(1 / k): the rounded product the stepper itself would publish, kept out of contraction.
            const auto &pl_ = loc[lc.parent];
            return "hy_mul_nc(jetw[" + std::to_string((k - 1u) * rowst + pl_.off) + "u + q * " + std::to_string(pl_.stride) + "u], "
                   + fp_literal(1. / static_cast<double>(k)) + ")";
        };
        // (With the exclusion test below the jets of the event equations are stored behind it, and only by wavefronts in which
        // an event is possible: the detection kernel does not read the jets of the systems the test has ruled out.)
        const bool ev_store_late = true;
        const std::function<std::string(std::uint32_t, std::uint32_t, const std::string &)> ev_store
            = [&](std::uint32_t ev, std::uint32_t k, const std::string &v) {
                  if (ev_store_late) {
                      return std::string{};
                  }
                  return "a.ev_tc[(u64)" + std::to_string(static_cast<std::uint64_t>(ev) * (order + 1u) + k) + "u * N + s] = " + v + ";\n";
              };
        // Terms of a sum side by side on the lanes of the system: leaf position p of the shared shape is read at a per-lane
        // offset (evo<p>: the jet column, of the variable itself or - position-type variables - of the variable it is
        // derived from; evz<p>: the current value of a position-type variable).
        ev_lane_hooks hooks;
        hooks.max_terms = std::min<std::uint32_t>(4u, L);
        hooks.sv_class = [&](std::uint32_t i) { return loc[i].derived ? 1 : 0; };
        hooks.sv_lane = [&](std::uint32_t pp, std::uint32_t k, int cls) -> std::string {
            const auto ps = std::to_string(pp);
            if (cls == 0) {
                return "jetw[" + std::to_string(k * rowst) + "u + evo" + ps + "]";
            }
            if (k == 0u) {
                return "jetw[evz" + ps + "]";
            }
            return "hy_mul_nc(jetw[" + std::to_string((k - 1u) * rowst) + "u + evo" + ps + "], " + fp_literal(1. / static_cast<double>(k)) + ")";
        };
        hooks.lane_bcast = [&](const std::string &v, std::uint32_t c) {
            return "__shfl(" + v + ", (int)((threadIdx.x & " + std::to_string(64u - L) + "u) + " + std::to_string(c) + "u), 64)";
        };
        std::string why_ev;
        const bool use_lanes = true;
        // Close encounters: an event equation |r_i - r_j|^2 + c is, up to the constant, the squared distance whose Taylor
        // coefficients the lane of the pair (i, j) holds as the history of its pow recurrence (sB: b_k / b_0). Such events are
        // not evaluated at all: the lane of the pair contributes its history - one event per lane, every lane at the same
        // time (one exclusion test, one set of stores for ALL of them); order p, which the recursion of the state does not
        // need, costs one more convolution from the order-p coefficients of the positions.
        std::vector<char> ev_on_lane(opts.ev_prog->ev_u.size(), 0);
        if (pairk && ev_store_late && opts.dev.pair_events) {
            for (std::size_t ev = 0; ev < opts.ev_prog->ev_u.size(); ++ev) {
                pair_distance_event pe;
                const bool pe_ok = match_pair_distance_event(*opts.ev_prog, opts.ev_prog->ev_u[ev], pe);
                if (!pe_ok) {
                    continue;
                }
                const auto unordered = [](std::pair<std::uint32_t, std::uint32_t> x) {
                    return x.first < x.second ? x : std::pair<std::uint32_t, std::uint32_t>{x.second, x.first};
                };
                std::vector<std::pair<std::uint32_t, std::uint32_t>> want;
                for (const auto &d : pe.diffs) {
                    want.push_back(unordered(d));
                }
                std::sort(want.begin(), want.end());
                for (std::uint32_t cl = 0; cl < nc && cl < L; ++cl) {
                    std::vector<std::pair<std::uint32_t, std::uint32_t>> have;
                    for (std::uint32_t i = 0; i < 3u; ++i) {
                        have.push_back(unordered({pl.ext_u[cl][pp.de[i][0]], pl.ext_u[cl][pp.de[i][1]]}));
                    }
                    std::sort(have.begin(), have.end());
                    if (have == want && pe_lane_ev.count(cl) == 0u) {
                        pe_lane_ev[cl] = {static_cast<std::uint32_t>(ev), pe.c};
                        pe_lane_sign[cl] = pe.sign;
                        ev_on_lane[ev] = 1;
                        break;
                    }
                }
            }
        }
        const bool ev_ok = emit_event_jets_inline(*opts.ev_prog, opts, sv, ev_store, ev_code, ev_coeffs, why_ev,
                                                  use_lanes ? &hooks : nullptr, &ev_on_lane);
        if (!ev_ok) {
            pe_lane_ev.clear();
        }
        if (ev_ok) {
            ev_inline = true;
            // Per-lane offsets of the leaf positions (lanes beyond the last term replicate term 0).
            const auto pick = [&](const std::vector<std::uint64_t> &v) {
                std::string r = std::to_string(v[0]) + "u";
                for (std::size_t c = v.size(); c-- > 1u;) {
                    r = "(l == " + std::to_string(c) + "u ? " + std::to_string(v[c]) + "u : " + r + ")";
                }
                return r;
            };
            for (std::size_t pp = 0; use_lanes && pp < hooks.leaf_vars.size(); ++pp) {
                std::vector<std::uint64_t> off, str_, zoff, zstr;
                for (const auto var : hooks.leaf_vars[pp]) {
                    const auto &lc = loc[var];
                    const auto &src_ = lc.derived ? loc[lc.parent] : lc;
                    off.push_back(src_.off);
                    str_.push_back(src_.stride);
                    zoff.push_back(lc.off);
                    zstr.push_back(lc.stride);
                }
                src << "const unsigned evo" << pp << " = " << pick(off) << " + q * " << pick(str_) << ";\n";
                if (hooks.leaf_class[pp] == 1) {
                    src << "const unsigned evz" << pp << " = " << pick(zoff) << " + q * " << pick(zstr) << ";\n";
                }
            }
            if (!pe_lane_ev.empty()) {
                // Per lane: 1 / 0 (the pair of this lane has an event), the constant of the event equation, the row block of
                // the event in a.ev_tc (lanes without an event: a block of their own behind the last event).
                std::string on = "0.0", cst = "0.0", row = std::to_string(opts.ev_prog->ev_u.size()) + "u";
                for (const auto &[lane_, evc] : pe_lane_ev) {
                    const auto ls = "(l == " + std::to_string(lane_) + "u ? ";
                    // (The switch carries the sign of the squared distance in the event equation: +-1, 0 = no event.)
                    on = ls + (pe_lane_sign.at(lane_) < 0 ? "-1.0 : " : "1.0 : ") + on + ")";
                    cst = ls + fp_literal(evc.second) + " : " + cst + ")";
                    row = ls + std::to_string(evc.first) + "u : " + row + ")";
                }
                src << "const double pe_on = " << on << ";\nconst double pe_c = " << cst << ";\nconst u64 pe_row = (u64)" << row
                    << " * " << (order + 1u) << "u;\n";
            }
        }
    }
    src << R"HIP(
// Work distribution. Propagation (steps per system differ): a device-side queue, one group of systems at a time (taking
// chunks of 8 groups per atomic costs 1 % there: consecutive groups no longer run at the same time on neighbouring
// wavefronts, which is what lets the partial-line accesses to the SoA arrays meet in L2). Single steps (equal cost per
// group): a static interleaved schedule - group = iteration * wavefronts + wavefront - with the same locality and no
// atomics (a launch of 1 048 576 systems is 524 288 atomics on one address: 3 ms of a 6.5 ms launch).
const u64 hy_waves = (u64)gridDim.x * HY_WPB;
const bool hy_static = (HY_MODE != 1) && (HY_NO_STATIC == 0);
u64 hy_it = 0;
bool hy_queue_empty = false;
for (;;) {
u64 base = 0;
if (hy_static) {
    // (Exit decided per workgroup: the cooperative store of the Taylor coefficients below has workgroup barriers. A
    // wavefront beyond the end of the ensemble in a live workgroup replicates the last system, without side effects.)
    if (HY_M4 && (hy_it * (u64)gridDim.x + blockIdx.x) * (HY_WPB * SPW) >= N) break;
    base = (hy_it * hy_waves + gwave) * SPW;
    ++hy_it;
} else {
    if (lane == 0u) base = atomicAdd((u64 *)(a.counters + 2), (u64)SPW);
}
// NOTE: through readfirstlane the position is a scalar for the compiler and the exit of the work loop a wave-uniform
// branch (a shuffle / the wavefront index leave it "divergent": exec-mask bookkeeping around the whole step loop).
base = ((u64)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) | (u64)__builtin_amdgcn_readfirstlane((unsigned)base);
if (!(HY_M4 && hy_static) && base >= N) break;
// NOTE: lanes beyond the end of the ensemble replicate the last system (no side effects).
bool live = (base + q) < N;
u64 s = live ? (base + q) : (N - 1u);
double t_hi = a.time_hi[s], t_lo = a.time_lo[s];
)HIP";
    for (std::uint32_t i = 0; i < p.n_par; ++i) {
        src << "const double par_" << i << " = a.pars[(u64)" << i << "u * N + s];\n";
    }
    for (const auto t : lane_par_tbls) {
        src << "const double lp" << t << " = a.pars[(u64)" << utname(t) << " * N + s];\n";
    }
    for (const auto &rg : rounds) {
        for (const auto &gr : rg) {
            for (const auto &ow : gr.owners) {
                src << row0_w(ow) << " = a.state[(u64)hy_utbl[" << ow.var_tbl * L << "u + l] * N + s];\n";
            }
        }
    }
    src << "HY_WSYNC();\n" << (jet_lds ? "" : "__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, \"wavefront\");\n");
    src << R"HIP(
hy_df tfin, rem;
tfin.hi = 0.0; tfin.lo = 0.0; rem.hi = 0.0; rem.lo = 0.0;
bool t_dir = true;
double mdt = __builtin_inf();
double step_lim = 0.0;
double thr = 0.0;
if (HY_MODE == 1) {
    tfin.hi = (a.tfin_hi != nullptr) ? a.tfin_hi[s] : a.tfin_s_hi;
    tfin.lo = (a.tfin_hi != nullptr) ? a.tfin_lo[s] : a.tfin_s_lo;
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    rem = hy_df_sub(tfin, tcur);
    t_dir = (rem.hi > 0.0) || (rem.hi == 0.0 && rem.lo >= 0.0);
    if (a.lim != nullptr) mdt = a.lim[s];
    // (Propagation from grid point to grid point - hy_kargs::pad bit 2 in mode 1, emitted_module::grid_multi_step: the
    // system leaves the step loop after the first step which reaches a.tc_thr[s], with the coefficients of that step stored.)
    if ((a.pad & 4) != 0) thr = a.tc_thr[s];
} else {
    step_lim = a.lim[s];
    // (Taylor coefficients of a lock-step sweep on demand - hy_kargs::pad bit 2, the loop of propagate_grid() without a
    // callback: the NEXT GRID TIME of every system - a.tc_thr, or a.tfin_hi for callers of before round 6 -, and only the
    // steps which reach it store their coefficients - see emitted_module::tc_by_threshold.)
    if ((a.pad & 4) != 0) thr = (a.tc_thr != nullptr) ? a.tc_thr[s] : a.tfin_hi[s];
}
u64 n_steps = 0, iter = 0;
double min_h = __builtin_inf(), max_h = 0.0, last_h = 0.0;
i64 outcome = HY_OC_SUCCESS;
// NOTE: the step loop is left by the whole wavefront at once (wave-uniform exit): a system which has reached its final
// time keeps executing steps of length zero with all its bookkeeping frozen by selects (fin) until the other
// systems of the wavefront are done - a wavefront executes an iteration as long as one of its lanes is active
// anyway. With a per-lane exit the loop-carried values of the lanes which have left live across several
// hundred live registers of the remaining ones, which is where this toolchain's live-range splitting goes
// wrong (DESIGN.md, toolchain notes).
bool fin = false;
// (Did the system leave the step loop through a step clamped to its remaining time? hy_kargs::grid_done.)
bool gfin = false;
int nf_seen = 0;
// (Stepper with events which evaluates the event equations itself: does this system need its Taylor coefficients in
// memory? See "On-demand Taylor coefficients" below. hy_kargs::pad: bit 0 = store them for every system, bit 1 = store
// NOTHING but them - the regeneration launch on the snapshot of the state before the step.)
bool need_tc = true;
const bool hy_tc_only = HY_M4 && ((a.pad & 2) != 0);
)HIP";
    // One-lane pair kernel: the per-system bookkeeping of the step loop (times, limits, counters, outcome: 30 registers
    // which nothing reads during the 20 orders) is parked in LDS between the tails of two steps - written at the end of
    // a tail, read back at the beginning of the next one - instead of being spilled to scratch by the register
    // allocator, whose reloads are scattered over the tail and each wait for a round trip through the vector memory
    // path. Every lane of a system holds the same values and stores them to the same address.
    const bool bk_lds = one_lane;
    const char *bk_fields_d[] = {"t_hi", "t_lo", "tfin.hi", "tfin.lo", "rem.hi", "rem.lo", "mdt", "step_lim", "min_h", "max_h", "last_h", "thr"};
    // (which = 0: every field; 1: the fields a step changes; 2: the others - final time and limits, which only change when a
    // system is picked up: an LDS store is the most expensive instruction of the kernel.)
    const auto bk_store = [&](int which = 0) {
        std::uint32_t f = 0;
        for (const auto *nm : bk_fields_d) {
            const std::string n_ = nm;
            const bool constant = n_ == "tfin.hi" || n_ == "tfin.lo" || n_ == "mdt" || n_ == "step_lim" || n_ == "thr";
            if (which == 0 || (which == 1) != constant) {
                src << "bk[" << f << "] = " << nm << ";\n";
            }
            ++f;
        }
        if (which == 2) {
            return;
        }
        src << "bk[" << f++ << "] = __longlong_as_double((long long)n_steps);\n";
        src << "bk[" << f++ << "] = __longlong_as_double((long long)iter);\n";
        src << "bk[" << f++ << "] = __longlong_as_double((long long)outcome);\n";
        src << "bk[" << f++ << "] = __longlong_as_double((long long)((t_dir ? 1 : 0) | (nf_seen != 0 ? 2 : 0) | (gfin ? 4 : 0)));\n";
    };
    const auto bk_load = [&]() {
        std::uint32_t f = 0;
        for (const auto *nm : bk_fields_d) {
            src << nm << " = bk[" << f++ << "];\n";
        }
        src << "n_steps = (u64)__double_as_longlong(bk[" << f++ << "]);\n";
        src << "iter = (u64)__double_as_longlong(bk[" << f++ << "]);\n";
        src << "outcome = (i64)__double_as_longlong(bk[" << f++ << "]);\n";
        src << "{\nconst long long fl = __double_as_longlong(bk[" << f++ << "]);\nt_dir = (fl & 1) != 0;\nnf_seen = (fl & 2) != 0 ? 1 : 0;\ngfin = (fl & 4) != 0;\n}\n";
    };
    if (bk_lds) {
        bk_store();
        src << "HY_WSYNC();\n";
    }
    src << "for (;;) {\n";
    src << body;
    // Close-encounter events on the lanes of their pairs (pe_lane_ev): g^[0] = b_0 + c, g^[k] = (b_k / b_0) b_0 from the
    // history of the pow recurrence, g^[p] from one more convolution; folded into the three norms of the lane BEFORE the
    // reduction over the lanes (masked: a lane without an event contributes nothing).
    std::vector<std::string> pe_g;
    if (!pe_lane_ev.empty()) {
        src << "HY_WSYNC();\n";
        std::string dK[3];
        for (std::uint32_t i = 0; i < 3u; ++i) {
            src << "const double pe_d" << i << " = " << slabk(order, utname(st1.s[i][0])) << " - " << slabk(order, utname(st1.s[i][1]))
                << ";\n";
            dK[i] = "pe_d" + std::to_string(i);
        }
        // b_p = 2 sum_i (sum_{j < p / 2} d_i[j] d_i[p - j] + 1/2 d_i[p / 2]^2)   (p even: src/detail/sum_sq.cpp:100-245)
        for (std::uint32_t i = 0; i < 3u; ++i) {
            src << "double pe_q" << i << " = " << dK[i] << " * " << sD[i][0] << ";\n";
            for (std::uint32_t j = 1; 2u * j < order; ++j) {
                src << "pe_q" << i << " = __builtin_fma(" << sD[i][j] << ", " << sD[i][order - j] << ", pe_q" << i << ");\n";
            }
            if (order % 2u == 0u) {
                src << "pe_q" << i << " = __builtin_fma(0.5 * " << sD[i][order / 2u] << ", " << sD[i][order / 2u] << ", pe_q" << i
                    << ");\n";
            }
        }
        src << "const double pe_bh = (pe_q0 + pe_q1) + pe_q2;\n";
        pe_g.resize(order + 1u);
        // (pe_on = +-1 on the lanes with an event: the sign of the squared distance in the event equation - exact. A lane
        // without an event gets zeros; it stores them into the spare block.)
        src << "const double pe_b0s = " << sB[0] << " * pe_on;\n";
        src << "const double pe_g0 = pe_b0s + pe_c;\n";
        pe_g[0] = "pe_g0";
        for (std::uint32_t k = 1; k < order; ++k) {
            src << "const double pe_g" << k << " = " << sB[k] << " * pe_b0s;\n";
            pe_g[k] = "pe_g" + std::to_string(k);
        }
        src << "const double pe_g" << order << " = (pe_bh + pe_bh) * pe_on;\n";
        pe_g[order] = "pe_g" + std::to_string(order);
        src << "m0 = hy_nmax(m0, fabs(pe_g0));\nmo = hy_nmax(mo, fabs(pe_g" << order
            << "));\nmom1 = hy_nmax(mom1, fabs(pe_g" << order - 1u << "));\n";
    }

    // Maximum over the lanes of the system: DPP stages where a DPP pattern yields an all-reduce step (xor 1, xor 2 within
    // quads; rotations by 4 and 8 within rows of 16 lanes once the quads are uniform), ds_bpermute for the others.
    // Packed tail (L >= 4): after the two stages inside the quads the three norms move to three lanes of every quad
    // (lane & 3 = 0: |x|, 1: |x^[p]|, 2 and 3: |x^[p-1]|) and the remaining stages reduce ONE value instead of three;
    // the logarithm of the selector is then evaluated once, on the packed lanes (hy_sel_log above).
    const bool packed_tail = L >= 4u;
    const auto red_ex = [&](std::uint32_t m, const char *v) -> std::string {
        const bool dpp_ok = true;
        if (dpp_ok && m == 1u) {
            return std::string("hy_dpp<0xB1>(") + v + ")";
        }
        if (dpp_ok && m == 2u) {
            return std::string("hy_dpp<0x4E>(") + v + ")";
        }
        if (dpp_ok && m == 4u && L % 16u == 0u) {
            return std::string("hy_dpp<0x124>(") + v + ")";
        }
        if (dpp_ok && m == 8u && L % 16u == 0u) {
            return std::string("hy_dpp<0x128>(") + v + ")";
        }
        return std::string("__shfl_xor(") + v + ", " + std::to_string(m) + ", 64)";
    };
    for (std::uint32_t m = 1; m < L; m *= 2u) {
        if (packed_tail && m == 4u) {
            break;
        }
        src << "m0 = hy_nmax(m0, " << red_ex(m, "m0") << ");\n";
        src << "mo = hy_nmax(mo, " << red_ex(m, "mo") << ");\n";
        src << "mom1 = hy_nmax(mom1, " << red_ex(m, "mom1") << ");\n";
    }
    if (packed_tail) {
        src << "const bool hy_q0 = (lane & 3u) == 0u, hy_q1 = (lane & 3u) == 1u;\n";
        src << "double nv = hy_q0 ? m0 : (hy_q1 ? mo : mom1);\n";
        for (std::uint32_t m = 4; m < L; m *= 2u) {
            src << "nv = hy_nmax(nv, " << red_ex(m, "nv") << ");\n";
        }
    }
    // Mode 4 (stepper with events): the norms over the state variables go to the kernel which extends them to the event
    // equations (hy_ev_jets). A wave-uniform branch; every lane of the system stores the same values.
    // (A separate specialisation of the kernel - opts.event_stepper - so that the propagation kernel is not touched: the
    // extra code, although never executed there, costs it spills in the step loop.)
    // Event equations inside the stepper: the statements prepared above (ev_code), then the three norms extended to them.
    if (ev_inline) {
        // (No scope around the statements: the coefficients of the event equations are read again by the exclusion test once
        // the step size is known. Their names live in a range of their own.)
        src << "double evm0 = 0.0, evmo = 0.0, evmom1 = 0.0;\n" << ev_code;
        for (const auto &c : ev_coeffs) {
            if (c.empty()) {
                continue;
            }
            src << "evm0 = hy_max(evm0, fabs(" << c[0] << "));\nevmo = hy_max(evmo, fabs(" << c[order]
                << "));\nevmom1 = hy_max(evmom1, fabs(" << c[order - 1u] << "));\n";
        }
        src << "nv = hy_nmax(nv, hy_q0 ? evm0 : (hy_q1 ? evmo : evmom1));\n";
        // (max |x_i| over the state variables and the event equations: the scale of the root finder's tolerance.)
        // (Not in the regeneration launch - hy_tc_only: it gets null pointers for everything but the coefficients.)
        src << "if (!hy_tc_only) a.max_abs_state[s] = hy_dpp<0x00>(nv);\n";
    }
    src << "#define HY_EV_INLINE " << (ev_inline ? 1 : 0) << "\n";
    src << "const bool nostate = " << ((m4 && !ev_inline) ? "true" : "false") << ";\n";
    // (The stepper with events never advances the time: hy_ev_post does, from the step size which was finally taken.)
    src << "const bool notime = " << (m4 ? "true" : "false") << ";\n";
    if (ev_inline) {
        // (Nothing to hand over: the norms are complete.)
    } else if (m4 && packed_tail) {
        // (Every lane stores: lane & 3 selects the row, the lanes of a system write identical values.)
        src << "a.sel_norms[(u64)(((lane & 3u) < 2u) ? (lane & 3u) : 2u) * N + s] = nv;\n";
    } else if (m4) {
        src << "a.sel_norms[s] = m0;\na.sel_norms[N + s] = mo;\na.sel_norms[2u * N + s] = mom1;\n";
    }
    // NOTE: rho = exp(log(x) / order) (hy_root): the minimum of the two estimates is taken on the exponents (exp is
    // monotone and keeps NaNs: the same selection as min(rho_o, rho_om1), src/taylor_02.cpp:1050-1072, one exp less).
    if (packed_tail) {
        // log(num / m) = log(num) - log(m): no quotients (0 -> +inf, inf -> -inf, inf - inf -> nan as for the quotient).
        src << "const double nw = (hy_q0 & (nv <= 1.0)) ? 1.0 : nv;\n";
        src << "const double lg = " << (sel_scalar ? "hy_sel_log_s" : "hy_sel_log") << "(nw);\n";
        src << "const double lg0 = hy_dpp<0x00>(lg), lg1 = hy_dpp<0x55>(lg), lg2 = hy_dpp<0xAA>(lg);\n";
        src << "const double lr_o = (lg0 - lg1) * " << fp_literal(1. / static_cast<double>(order)) << ";\n";
        src << "const double lr_om1 = (lg0 - lg2) * " << fp_literal(1. / static_cast<double>(order - 1u)) << ";\n";
        src << "const double rho_m = " << (sel_scalar ? "hy_sel_exp_s" : "exp") << "(hy_min(lr_o, lr_om1));\n";
    } else {
        src << "const double num_rho = (m0 <= 1.0) ? 1.0 : m0;\n";
        src << "const double lr_o = log(num_rho / mo) * " << fp_literal(1. / static_cast<double>(order)) << ";\n";
        src << "const double lr_om1 = log(num_rho / mom1) * " << fp_literal(1. / static_cast<double>(order - 1u)) << ";\n";
        src << "const double rho_m = exp(hy_min(lr_o, lr_om1));\n";
    }
    if (bk_lds) {
        bk_load();
    }
    // The limit of this step (remaining time / maximum step). NOTE: computed here, next to its only use, and not at the
    // top of the step: its inputs then do not have to be live (or reloaded) before the 20 orders.
    src << R"HIP(
double lim;
if (HY_MODE == 1) {
    hy_df m; m.lo = 0.0;
    // NOTE: selects, not an if/else on the (per-lane) direction: see the note on HY_LIBM1.
    m.hi = t_dir ? mdt : -mdt;
    const bool lt_fwd = hy_df_lt(rem, m), lt_bwd = hy_df_lt(m, rem);
    const bool rem_first = (t_dir & lt_fwd) | (!t_dir & lt_bwd);
    lim = rem_first ? rem.hi : m.hi;
} else {
    lim = step_lim;
}
lim = fin ? 0.0 : lim;
)HIP";
    src << "double h = rho_m * " << fp_literal(rhofac(order)) << ";\n";
    src << "h = hy_min(h, fabs(lim));\nh = (lim < 0.0) ? -h : h;\n";
    // A system which is done takes steps of length EXACTLY zero (lim = 0 gives that unless the selector produced a nan):
    // the double-length time, the remaining time, the state and the step counters then reproduce themselves bit by bit,
    // and only the values which a zero-length step would overwrite need a select below (last_h, outcome).
    src << "h = fin ? 0.0 : h;\n";
    if (ev_inline) {
        // On-demand Taylor coefficients. Behind a step with events nothing reads the coefficients of the state variables
        // unless an event is detected (its callback may ask for dense output) or the caller asks for them. The stepper
        // runs the fast exclusion test of the detection (interval Horner enclosure of the event polynomial over the
        // step, src/detail/event_detection.cpp:704-816) on the jets it has just computed - CONSERVATIVELY: a system counts
        // as event-free only if the enclosure stays away from zero by 1e-8 of the largest intermediate magnitude, six
        // orders of magnitude above the rounding differences between this evaluation and the detection kernel's own -
        // and the workgroup stores its coefficients only if one of its systems may have an event. The integrator keeps
        // a snapshot of the state before the step: whoever reads coefficients which were not stored gets them from a
        // second launch on the snapshot (pad = 3), bit-identical.
        // First a cheaper bound which decides almost every system: |P(t) - c_0| <= sum_k |c_k| |h|^k on the step (20
        // multiply-adds per event equation); the interval enclosure (300 instructions per event equation) runs - behind a
        // wave-uniform branch - only in wavefronts where it leaves a system undecided.
        src << "bool maybe0 = false, pe_m0 = false;\n{\nconst double ah = fabs(h);\n";
        for (const auto &c : ev_coeffs) {
            if (c.empty()) {
                continue;
            }
            src << "{\ndouble r = fabs(" << c[order] << ");\n";
            for (std::uint32_t k = order - 1u; k >= 1u; --k) {
                src << "r = r * ah + fabs(" << c[k] << ");\n";
            }
            src << "r = r * ah;\nmaybe0 = maybe0 | !(fabs(" << c[0] << ") > r * 1.00000001);\n}\n";
        }
        // (The events on the lanes of their pairs: every lane tests ITS event - one instruction stream for all of them.)
        if (!pe_g.empty()) {
            src << "{\ndouble r = fabs(" << pe_g[order] << ");\n";
            for (std::uint32_t k = order - 1u; k >= 1u; --k) {
                src << "r = r * ah + fabs(" << pe_g[k] << ");\n";
            }
            src << "r = r * ah;\npe_m0 = (pe_on != 0.0) & !(fabs(" << pe_g[0] << ") > r * 1.00000001);\n}\n";
        }
        src << "}\nneed_tc = ((a.pad & 1) != 0);\nbool ev_possible = false;\n";
        // (Any lane of the system: one ballot, then the bits of the system's lanes.)
        const auto sys_any = [&](const char *v) {
            return "(((__builtin_amdgcn_ballot_w64(" + std::string(v) + ") >> ((threadIdx.x & 63u) & " + std::to_string(64u - L) + "u)) & "
                   + std::to_string((std::uint64_t(1) << L) - 1u) + "ull) != 0ull)";
        };
        src << "if (__builtin_amdgcn_ballot_w64(maybe0 | pe_m0) != 0ull) {\n";
        src << "bool maybe = false, maybe_te = false;\nconst double lo_h = (h < 0.0) ? h : 0.0, hi_h = (h < 0.0) ? 0.0 : h;\n";
        for (std::size_t ci = 0; ci < ev_coeffs.size(); ++ci) {
            const auto &c = ev_coeffs[ci];
            if (c.empty()) {
                continue;
            }
            src << "{\ndouble lo = " << c[order] << ", hi = lo, mm = fabs(lo);\n";
            for (std::uint32_t i = 1; i <= order; ++i) {
                src << "{\nconst double p0 = lo * lo_h, p1 = lo * hi_h, p2 = hi * lo_h, p3 = hi * hi_h;\n"
                    << "const double mn = fmin(fmin(p0, p1), fmin(p2, p3)), mx = fmax(fmax(p0, p1), fmax(p2, p3));\n"
                    << "lo = mn + " << c[order - i] << ";\nhi = mx + " << c[order - i] << ";\n"
                    << "mm = fmax(mm, fmax(fabs(lo), fabs(hi)));\n}\n";
            }
            src << "const bool excl = (((lo > 0.0) & (hi > 0.0)) | ((lo < 0.0) & (hi < 0.0))) & (fmin(fabs(lo), fabs(hi)) > 1e-8 * mm);\n"
                << "maybe = maybe | !excl;\n" << (ci < opts.n_t_events ? "maybe_te = maybe_te | !excl;\n" : "") << "}\n";
        }
        src << "bool pe_m = false;\n";
        if (!pe_g.empty()) {
            src << "{\ndouble lo = " << pe_g[order] << ", hi = lo, mm = fabs(lo);\n";
            for (std::uint32_t i = 1; i <= order; ++i) {
                src << "{\nconst double p0 = lo * lo_h, p1 = lo * hi_h, p2 = hi * lo_h, p3 = hi * hi_h;\n"
                    << "const double mn = fmin(fmin(p0, p1), fmin(p2, p3)), mx = fmax(fmax(p0, p1), fmax(p2, p3));\n"
                    << "lo = mn + " << pe_g[order - i] << ";\nhi = mx + " << pe_g[order - i] << ";\n"
                    << "mm = fmax(mm, fmax(fabs(lo), fabs(hi)));\n}\n";
            }
            src << "const bool excl = (((lo > 0.0) & (hi > 0.0)) | ((lo < 0.0) & (hi < 0.0))) & (fmin(fabs(lo), fabs(hi)) > 1e-8 * mm);\n"
                << "pe_m = pe_m0 & !excl;\n}\n";
        }
        // (The coefficients of the state variables are read behind the stepper only where a step may be TRUNCATED: at a
        // TERMINAL event: the events on the lanes of their pairs count as terminal whenever the integrator has one.)
        src << "ev_possible = (maybe & maybe0) | " << sys_any("pe_m") << ";\n";
        if (opts.n_t_events != 0u) {
            src << "need_tc = need_tc | (maybe_te & maybe0) | " << sys_any("pe_m") << ";\n";
        }
        src << "}\n";
        // (For the detection kernel: systems in which no event is possible are skipped without reading their event jets -
        // which are stored only by the wavefronts that hold such a system.)
        src << "if (!hy_tc_only) a.sel_norms[s] = ev_possible ? 1.0 : 0.0;\n";
        // (A caller who asked for the coefficients of every system - the raw stepper ABI, loops with dense output - gets the
        // event jets of every system as well.)
        src << "if (!hy_tc_only && (((a.pad & 1) != 0) | (__builtin_amdgcn_ballot_w64(ev_possible) != 0ull))) {\n";
        for (std::size_t ev = 0; ev < ev_coeffs.size(); ++ev) {
            for (std::uint32_t k = 0; !ev_coeffs[ev].empty() && k <= order; ++k) {
                src << "a.ev_tc[(u64)" << static_cast<std::uint64_t>(ev) * (order + 1u) + k << "u * N + s] = " << ev_coeffs[ev][k] << ";\n";
            }
        }
        // (The lanes of the pairs: every lane stores the row of ITS event - lanes without one into the spare block.)
        for (std::uint32_t k = 0; !pe_g.empty() && k <= order; ++k) {
            src << "a.ev_tc[(pe_row + " << k << "u) * N + s] = " << pe_g[k] << ";\n";
        }
        src << "}\n";
    } else if (ev_inline) {
        src << "if (!hy_tc_only) a.sel_norms[s] = 1.0;\n";
    }

    src << "asm volatile(\"\" ::: \"memory\");\n";
    // NOTE: with the jets in global scratch the lanes exchange them through memory: wavefront-scope fences.
    const char *jet_fence = jet_lds ? "" : "__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, \"wavefront\");\n";
    src << jet_fence;
    // Final evaluation of the Taylor series of the state variables (taylor_run_multihorner() / taylor_run_ceval(),
    // src/taylor_00.cpp:279-460). The jets live in LDS (or in the per-wave scratch), so the evaluation is spread
    // evenly over the lanes of the group: slot h, lane l <-> column h * L + l, ceil(n_eq / L) slots instead of
    // one per owner slot (3 instead of 4 for the 36 variables of the outer Solar System on 16 lanes).
    const auto kstride = static_cast<std::uint64_t>(spw) * n_colp;
    // (name of the new value, where the current value is read, where the new one goes)
    std::vector<std::tuple<std::string, std::string, std::string>> upd;
    if (one_lane) {
        // One pass per glue round: the variable with a jet column (v) and the one derived from it (x' = v), whose
        // coefficients are formed on the fly from the same column, x^[k] = v^[k-1] * RN(1 / k) - bit for bit the published
        // coefficient (ssa_emitter::div_const() in its reciprocal form). Both sums of a lane share the loads of the column
        // (one LDS read per order) and the powers of h; every pass is one instruction stream without selects.
        for (const auto &rg : rounds) {
            for (const auto &gr : rg) {
                for (const auto &ow : gr.owners) {
                    if (ow.derived) {
                        continue;
                    }
                    // The variables derived from this one (at most one: chains of length <= 2).
                    const owner_slot *dv = nullptr;
                    for (const auto &o2 : gr.owners) {
                        if (o2.derived && o2.parent == ow.col) {
                            dv = &o2;
                        }
                    }
                    const auto xn = "xn" + std::to_string(ow.col);
                    const auto xd = dv != nullptr ? "xn" + std::to_string(dv->col) : std::string{};
                    const auto colp = "jr" + std::to_string(ow.col);
                    if (dv != nullptr && pk_tbl.count(ow.col) != 0u) {
                        // A partially filled owner slot (18 velocity columns on 16 lanes leave 2): ONE series per lane - the
                        // lanes [0, n) sum the velocity columns, the lanes [n, 2 n) the series derived from them, whose
                        // coefficient k is row k - 1 of the same column times RN(1 / k); both kinds run the same statements, the
                        // row shift sits in the lane's pointer and the factor (1 or RN(1 / k)) comes from a two-row table in LDS.
                        // Six instructions per order instead of the ten of the two-series pass which 2 of 16 lanes used.
                        const auto cs = std::to_string(ow.col);
                        const auto &tb = pk_tbl.at(ow.col);
                        // (The current value of the lane's variable: order-0 row of the column / entry of the derived variable.)
                        src << "double *const pk_v" << cs << " = jetw + q * " << jet_sys(ow) << "u + hy_utbl[" << tb[0] * L << "u + l];\n";
                        src << "double " << xn << ";\n{\n";
                        // (Row k of the lane's series at pk_j[(k - 1) * stride]: the velocity column from row 1 on, from row 0
                        // on for the derived series.)
                        src << "const double *const pk_j = jetw + q * " << jet_sys(ow) << "u + hy_utbl[" << tb[1] * L << "u + l];\n";
                        src << "const double *const pk_f = lds_fac + hy_utbl[" << tb[2] * L << "u + l];\n";
                        src << "double res = pk_v" << cs << "[0], comp = 0.0, cur_h = h;\n";
                        for (std::uint32_t k = 1; k <= order; ++k) {
                            src << "{\nconst double ck = pk_j[" << (k - 1u) * kstride << "] * pk_f[" << k << "];\n"
                                << "const double tmp = ck * cur_h;\nconst double y = tmp - comp;\nconst double t = res + y;\n"
                                << "comp = (t - res) - y;\nres = t;\n";
                            if (k < order) {
                                src << "cur_h = cur_h * h;\n";
                            }
                            src << "}\n";
                        }
                        src << xn << " = res;\n}\n";
                        upd.emplace_back(xn, "pk_v" + cs + "[0]", "pk_v" + cs + "[0]");
                        continue;
                    }
                    src << "double " << xn << ";\n";
                    if (dv != nullptr) {
                        src << "double " << xd << ";\n";
                    }
                    src << "{\n";
                    if (opts.high_accuracy) {
                        src << "double cprev = " << colp << "[0];\ndouble res = cprev, comp = 0.0, cur_h = h;\n";
                        if (dv != nullptr) {
                            src << "double resd = " << row0_r(*dv) << ", compd = 0.0;\n";
                        }
                        for (std::uint32_t k = 1; k <= order; ++k) {
                            src << "{\nconst double ck = " << colp << "[" << k * kstride << "];\n";
                            if (dv != nullptr) {
                                src << "const double cd = cprev * " << fp_literal(1. / static_cast<double>(k)) << ";\n"
                                    << "const double tmpd = cd * cur_h;\nconst double yd = tmpd - compd;\n"
                                    << "const double td = resd + yd;\ncompd = (td - resd) - yd;\nresd = td;\n";
                            }
                            src << "const double tmp = ck * cur_h;\nconst double y = tmp - comp;\nconst double t = res + y;\n"
                                << "comp = (t - res) - y;\nres = t;\ncprev = ck;\n";
                            if (k < order) {
                                src << "cur_h = cur_h * h;\n";
                            }
                            src << "}\n";
                        }
                    } else {
                        // Horner from the highest order down: the derived series needs the column shifted by one.
                        src << "double res = " << colp << "[" << order * kstride << "];\n";
                        if (dv != nullptr) {
                            src << "double resd = " << colp << "[" << (order - 1u) * kstride << "] * "
                                << fp_literal(1. / static_cast<double>(order)) << ";\n";
                        }
                        for (std::uint32_t k = 1; k <= order; ++k) {
                            const auto kk = order - k;
                            src << "res = " << colp << "[" << kk * kstride << "] + res * h;\n";
                        }
                        if (dv != nullptr) {
                            for (std::uint32_t k = 1; k <= order; ++k) {
                                const auto kk = order - k;
                                if (kk >= 1u) {
                                    src << "resd = (" << colp << "[" << (kk - 1u) * kstride << "] * "
                                        << fp_literal(1. / static_cast<double>(kk)) << ") + resd * h;\n";
                                } else {
                                    src << "resd = " << row0_r(*dv) << " + resd * h;\n";
                                }
                            }
                        }
                    }
                    src << xn << " = res;\n";
                    if (dv != nullptr) {
                        src << xd << " = resd;\n";
                    }
                    src << "}\n";
                    // (Event equations inside the stepper: the order-0 rows stay what they were - the Taylor coefficients
                    // leave through the cooperative store behind the step - and the new state waits in the slab, which is
                    // dead between the last order and the next step.)
                    const auto new_at = [&](const owner_slot &o) {
                        return ev_inline ? ("slab[" + std::to_string(o.col * L) + "u + l]") : row0_w(o);
                    };
                    upd.emplace_back(xn, row0_r(ow), new_at(ow));
                    if (dv != nullptr) {
                        upd.emplace_back(xd, row0_r(*dv), new_at(*dv));
                    }
                }
            }
        }
    }
    for (std::uint32_t c = 0; !one_lane && c < n_hslots; ++c) {
        upd.emplace_back("xn" + std::to_string(c), "hc" + std::to_string(c) + "[0]", "hc" + std::to_string(c) + "[0]");
        if (m4) {
            // (No state update in the stepper with events.)
            src << "const double xn" << c << " = hc" << c << "[0];\n";
            continue;
        }
        src << "double xn" << c << ";\n{\nconst double *c = hc" << c << ";\n";
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
        src << "xn" << c << " = res;\n}\n";
    }
    src << R"HIP(
double nt_hi, nt_lo;
{
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    hy_df hh; hh.hi = h; hh.lo = 0.0;
    const hy_df nt = hy_df_add(tcur, hh);
    nt_hi = nt.hi; nt_lo = nt.lo;
}
int nfi = !(hy_finite(nt_hi) && hy_finite(nt_lo)) ? 1 : 0;
)HIP";
    for (const auto &u3 : upd) {
        // (The dummy column holds finite copies.)
        src << "nfi |= !hy_finite(" << std::get<0>(u3) << ") ? 1 : 0;\n";
    }
    // Any lane of the system: one ballot, then the bits of the system's lanes.
    src << "{\nconst u64 nfb = __builtin_amdgcn_ballot_w64(nfi != 0);\n";
    if (L == 64u) {
        src << "nfi = (nfb != 0ull) ? 1 : 0;\n}\n";
    } else {
        src << "nfi = (((nfb >> ((threadIdx.x & 63u) & " << (64u - L) << "u)) & " << ((std::uint64_t(1) << L) - 1u)
            << "ull) != 0ull) ? 1 : 0;\n}\n";
    }
    // Taylor coefficients on request (wave-uniform branch). NOTE: every lane stores: the idle lanes of a partially filled
    // owner slot replicate the variable of a valid lane and the lanes beyond the end of the ensemble replicate the
    // last system, so their stores write the same values to the same addresses - no divergent region in the step
    // loop (see the toolchain notes in DESIGN.md). A rolled loop with a running pointer: unrolled, the (order + 1)
    // store addresses per owner slot are invariants of the step loop and get hoisted into registers (84 x 64 bit for
    // the outer Solar System) for a path that only runs when the caller asks for the coefficients.
    if (one_lane && jet_lds && !m4) {
        // (A zero-length step - a system which has reached its last grid point - stores like the reference's lock-step
        // loop does; a non-finite new time compares false and stores nothing: nobody reads those coefficients.)
        // A step clamped to its limit stores as well: the last step of a lane ends at tlast - rem.lo, whose high part can
        // land on either side of the last grid time (grids ending at 0, crossing 0, backward runs to 0) while hy_grid_post
        // (h == rem.hi) evaluates every remaining grid point from these coefficients. (Steps clamped by max_delta_t store
        // needlessly: harmless.)
        src << "const bool hy_reach = (h > 0.0) ? (nt_hi >= thr) : (nt_hi <= thr);\n";
        // (fin: a system which left the step loop in an EARLIER iteration idles with zero-length steps until its wavefront is
        // done or its slot is refilled - in a launch from grid point to grid point those must not overwrite the coefficients
        // of its exit step.)
        src << "const bool tc_sys = ((a.pad & 4) == 0) | (!fin & ((h == 0.0) | (h == lim) | hy_reach));\n";
        src << "if (a.tc != nullptr && tc_sys) {\n";
    } else {
        src << (jet_lds ? "if (a.tc != nullptr && !HY_M4) {\n" : "if (a.tc != nullptr) {\n");
    }
    for (const auto &rg : rounds) {
        for (const auto &gr : rg) {
            for (const auto &ow : gr.owners) {
                if (ow.derived) {
                    src << "{\nconst double *c = jr" << ow.parent << ";\ndouble *tcp = a.tc + ((u64)hy_utbl[" << ow.var_tbl * L
                        << "u + l] * " << (order + 1u) << "u) * N + s;\n*tcp = " << row0_r(ow) << ";\ntcp += N;\n"
                        << "#pragma nounroll\nfor (unsigned k = 1; k <= " << order << "u; ++k) {\n*tcp = c[(u64)(k - 1u) * "
                        << kstride << "u] * hy_rk[k];\ntcp += N;\n}\n}\n";
                    continue;
                }
                src << "{\nconst double *c = jr" << ow.col << ";\ndouble *tcp = a.tc + ((u64)hy_utbl[" << ow.var_tbl * L
                    << "u + l] * " << (order + 1u) << "u) * N + s;\n"
                    << "#pragma nounroll\nfor (unsigned k = 0; k <= " << order << "u; ++k) {\n*tcp = c[(u64)k * "
                    << kstride << "u];\ntcp += N;\n}\n}\n";
            }
        }
    }
    src << "}\n";
    // The new state becomes the order-0 row of the jets (read back by the owner lanes at the top of the next step).
    src << "HY_WSYNC();\n" << jet_fence;
    for (const auto &u3 : upd) {
        // (A zero-length step leaves the state untouched bit by bit: x + 0 * ... = x also in the compensated sum.)
        src << std::get<2>(u3) << " = (fin | nostate) ? " << std::get<1>(u3) << " : " << std::get<0>(u3) << ";\n";
    }
    src << "HY_WSYNC();\n" << jet_fence;
    src << R"HIP(
{
    // Bookkeeping of the reference's step() / propagate_until() loops (src/taylor_adaptive_batch.cpp:632-727,
    // :1395-1520) on values frozen once the system is done.
    const bool nf = nfi != 0;
    const i64 oc_new = nf ? HY_OC_ERR_NF_STATE : ((h == lim) ? HY_OC_TIME_LIMIT : HY_OC_SUCCESS);
    bool done = nf | (HY_MODE != 1);
    const u64 ns_new = n_steps + ((!done & (h != 0.0)) ? 1u : 0u);
    const bool upd = !done & (oc_new == HY_OC_SUCCESS);
    const double ah = fabs(h);
    const double mn_new = upd ? hy_min(min_h, ah) : min_h;
    const double mx_new = upd ? hy_max(max_h, ah) : max_h;
    done |= (h == rem.hi);
    gfin = fin ? gfin : (h == rem.hi);
    done |= HY_GRID_STOP;
    hy_df tnew; tnew.hi = nt_hi; tnew.lo = nt_lo;
    const hy_df rem_new = hy_df_sub(tfin, tnew);
    const u64 it_new = iter + 1u;
    const bool sl = !done & (it_new == a.max_steps);
    const i64 oc_fin = sl ? HY_OC_STEP_LIMIT : oc_new;
    done |= sl;
    nf_seen |= (!fin & nf & !nostate) ? 1 : 0;
    // (fin: h = 0, hence nt = t, rem_new = rem, ns_new = n_steps, no min / max update - see above.)
    t_hi = notime ? t_hi : nt_hi;
    t_lo = notime ? t_lo : nt_lo;
    last_h = fin ? last_h : h;
    outcome = fin ? outcome : oc_fin;
    n_steps = ns_new;
    min_h = mn_new;
    max_h = mx_new;
    rem.hi = rem_new.hi;
    rem.lo = rem_new.lo;
    iter = it_new;
    fin = fin | done;
}
)HIP";
    // Per-system refill (propagations through the device-side work queue): the step loop is left by the whole wavefront at
    // once, so a system which reaches its final time early idles - zero-length steps - until the slowest of the SPW
    // systems of its wavefront is done. With heterogeneous step counts (per-lane final times T * U(0.5, 1.5): 37 .. 127 steps
    // per system) that costs E[max of 4] / mean = 1.3x: 0.78 of the rate of a uniform ensemble (bench.py, divergence leg).
    // Here a finished system is retired on the spot - its results stored like at the end of the kernel - and its lanes pull
    // the next system from the queue: one atomic per system instead of one per wavefront, only behind a wave-uniform test
    // that some system of the wavefront has just finished. (The reference's lanes idle like the old loop:
    // src/taylor_adaptive_batch.cpp:1378-1460.)
    const bool refill = one_lane && !m4 && jet_lds && p.n_par == 0u && lane_par_tbls.empty() && L < 64u
                        && opts.dev.refill;
    if (refill) {
        src << "if (!hy_static && !hy_queue_empty && __builtin_amdgcn_ballot_w64(fin) != 0ull) {\n";
        // 1. Retire.
        for (const auto &rg : rounds) {
            for (const auto &gr : rg) {
                for (const auto &ow : gr.owners) {
                    src << "if (fin && ovalid" << ow.col << " && live) a.state[(u64)hy_utbl[" << ow.var_tbl * L
                        << "u + l] * N + s] = " << row0_w(ow) << ";\n";
                }
            }
        }
        src << R"HIP(
if (fin && l == 0u && live) {
    a.time_hi[s] = t_hi;
    a.time_lo[s] = t_lo;
    a.last_h[s] = last_h;
    a.outcome[s] = outcome;
    a.min_h[s] = min_h;
    a.max_h[s] = max_h;
    a.n_steps[s] = n_steps;
    if (HY_GRID_FLAGS) a.grid_done[s] = gfin ? 1.0 : 0.0;
    if (nf_seen != 0) atomicAdd(a.counters, 1u);
}
// 2. The next system of the queue, for every finished system of the wavefront (its lane 0 asks, the others listen).
u64 snew = 0;
if (fin && l == 0u) snew = atomicAdd((u64 *)(a.counters + 2), (u64)1);
{
    const int src_lane = (int)((threadIdx.x & 63u) - l);
    const unsigned lo_ = (unsigned)__shfl((int)(unsigned)snew, src_lane, 64);
    const unsigned hi_ = (unsigned)__shfl((int)(unsigned)(snew >> 32), src_lane, 64);
    snew = ((u64)hi_ << 32) | (u64)lo_;
}
const bool got = fin && (snew < N);
// (The queue position only grows: one request beyond the end means that the queue is empty - stop asking.)
if (__builtin_amdgcn_ballot_w64(fin && !got) != 0ull) hy_queue_empty = true;
live = fin ? got : live;
nf_seen = fin ? 0 : nf_seen;
s = got ? snew : s;
// 3. Its state, times and limits; a fresh set of counters.
if (got) {
    t_hi = a.time_hi[s];
    t_lo = a.time_lo[s];
)HIP";
        for (const auto &rg : rounds) {
            for (const auto &gr : rg) {
                for (const auto &ow : gr.owners) {
                    src << row0_w(ow) << " = a.state[(u64)hy_utbl[" << ow.var_tbl * L << "u + l] * N + s];\n";
                }
            }
        }
        src << R"HIP(
    tfin.hi = (a.tfin_hi != nullptr) ? a.tfin_hi[s] : a.tfin_s_hi;
    tfin.lo = (a.tfin_hi != nullptr) ? a.tfin_lo[s] : a.tfin_s_lo;
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    rem = hy_df_sub(tfin, tcur);
    t_dir = (rem.hi > 0.0) || (rem.hi == 0.0 && rem.lo >= 0.0);
    mdt = (a.lim != nullptr) ? a.lim[s] : __builtin_inf();
    thr = ((a.pad & 4) != 0) ? a.tc_thr[s] : 0.0;
    n_steps = 0;
    iter = 0;
    min_h = __builtin_inf();
    max_h = 0.0;
    last_h = 0.0;
    outcome = HY_OC_SUCCESS;
    fin = false;
    gfin = false;
}
HY_WSYNC();
)HIP";
        if (bk_lds) {
            bk_store(2);
        }
        src << "}\n";
    }
    if (bk_lds) {
        bk_store(1);
    }
    src << R"HIP(
if (__builtin_amdgcn_ballot_w64(!fin) == 0ull) break;
}
if (nf_seen != 0 && l == 0u && live) atomicAdd(a.counters, 1u);
)HIP";
    if (jet_lds && m4) {
        // Mode 4: the Taylor coefficients of the workgroup's systems (consecutive under the static schedule) leave through
        // a cooperative store - [variable][order] rows of HY_WPB * SPW consecutive systems, whole 128-byte lines for the
        // 16 systems of the lane-pair kernel - instead of 16-byte pieces per wavefront (the per-wavefront stores make
        // the 1 048 576-system stepper with events transaction bound: 8 ms instead of 3).
        const auto spb = wpb * spw;
        // NOTE: the (row, source offset, stride) of every stored row comes from a table in LDS (lds_tcsrc, filled once per
        // workgroup from hy_tc_src): with the table in global / constant memory every iteration of this loop issues vector
        // loads BEHIND the stores of the previous one, and gfx9 counts loads and stores in one in-order counter (vmcnt) -
        // each iteration then waits for a store acknowledgement (12 iterations: 6 us per group of systems, 1.5 ms of a
        // 4.2 ms launch on 1 048 576 systems).
        // (Event equations inside the stepper: only if one of the systems of the workgroup needs them - need_tc.)
        src << "{\nconst int hy_wg_tc = __syncthreads_or(need_tc ? 1 : 0);\n";
        src << "if (hy_wg_tc == 0 && threadIdx.x == 0u) atomicAdd(a.counters + 4, 1u);\n";
        src << "if (hy_wg_tc != 0) {\n";
        src << "const u64 bs0 = base - (u64)wib * SPW;\n";
        src << "for (unsigned idx = threadIdx.x; idx < " << n_tc_rows * spb << "u; idx += " << bs << "u) {\n";
        src << "const unsigned sy = idx % " << spb << "u;\nconst unsigned long long te = lds_tcsrc[idx / " << spb << "u];\n";
        src << "const unsigned row = (unsigned)(te & 0xfffffull), off = (unsigned)((te >> 20) & 0xfffffull), sst = "
               "(unsigned)(te >> 40);\n";
        src << "const u64 sg = bs0 + sy;\n";
        src << "const double val = lds_jet[(sy / SPW) * " << jet_doubles_per_wave << "u + off + (sy % SPW) * sst];\n";
        src << "if (sg < N) a.tc[(u64)row * N + sg] = val;\n}\n}\n__syncthreads();\n}\n";
    }
    src << R"HIP(
)HIP";
    for (const auto &rg : rounds) {
        for (const auto &gr : rg) {
            for (const auto &ow : gr.owners) {
                if (m4 && !ev_inline) {
                    // (The stepper with events leaves the state, the time and the step size to the kernels behind it -
                    // unless it evaluates the event equations itself: then the state is final here.)
                    continue;
                }
                src << "if (ovalid" << ow.col << " && live && !hy_tc_only) a.state[(u64)hy_utbl[" << ow.var_tbl * L << "u + l] * N + s] = "
                    << (ev_inline ? ("slab[" + std::to_string(ow.col * L) + "u + l]") : row0_w(ow)) << ";\n";
            }
        }
    }
    src << R"HIP(
if (l == 0u && live && !hy_tc_only) {
    if (!HY_M4) {
        if (HY_MODE != 2) {
            a.time_hi[s] = t_hi;
            a.time_lo[s] = t_lo;
        } else {
            const_cast<double *>(a.lim)[s] = last_h;
        }
    }
    if (!HY_M4 || HY_EV_INLINE) a.last_h[s] = last_h;
    a.outcome[s] = outcome;
    if (HY_MODE == 1) {
        a.min_h[s] = min_h;
        a.max_h[s] = max_h;
        a.n_steps[s] = n_steps;
        if (HY_GRID_FLAGS) a.grid_done[s] = gfin ? 1.0 : 0.0;
    }
}
}
}
)HIP";

    ret.source = src.str();
    ret.kernel_name = "hy_taylor";
    ret.dout_name = "hy_dout";
    ret.block_size = bs;
    ret.lanes_per_system = L;
    ret.lds_bytes = 0;
    ret.mode = emit_mode::cluster;
    ret.n_statements = e.n_stmt;
    ret.scratch_per_wave = jet_lds ? 0u : jet_doubles_per_wave;
    ret.persistent = true;
    if (pairk) {
        // NOTE: MachineLICM hoists the materialisation of ~50 fp64 literals (1 / k, the polynomial constants of the
        // step-size selector) out of the step loop into SGPR pairs: pointers and masks are then spilled to VGPR lanes and
        // come back through 186 v_readlane_b32 per step (VALU issue slots). Without it: 20, and 7 % fewer VALU
        // instructions in the loop (measured: +2 % system-steps/s, profiles/experiments/run17.sh).
        ret.compile_flags = "-mllvm -disable-machine-licm";
        if (one_lane) {
            // NOTE: no merging of LDS accesses: on gfx950 a ds_read2_b64 is serviced at half the rate of two ds_read_b64
            // (8 against 2 x 2 LDS cycles per wavefront) and this kernel is within 25 % of the LDS throughput.
            ret.compile_flags += " -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-load-store-vectorizer=0";
        }
    }
    ret.tc_optional = true;
    ret.cluster_mode4 = m4;
    ret.events_in_stepper = ev_inline;
    ret.compact_tc = compact_tc;
    ret.tc_by_threshold = one_lane && jet_lds && !m4;
    ret.grid_multi_step = ret.tc_by_threshold;
    one_lane_jets_in_lds = one_lane && jet_lds;
    ret.notes = std::string(one_lane ? "cluster mode v5 (one lane per pair, 2 wavefronts per SIMD): "
                                     : (pair_split ? "cluster mode v3 (lane pairs, 2 wavefronts per SIMD): " : "cluster mode v2 (pipelined): "))
                + std::to_string(nc) + " clusters of " + std::to_string(t0.size())
                + " nodes, L=" + std::to_string(L) + ", " + std::to_string(pl.n_slots) + " LDS slots x2, "
                + std::to_string(n_own) + " state-variable owner slots, " + std::to_string(utbl.size())
                + " slot tables, jets in " + (jet_lds ? "LDS" : "global scratch")
                + (one_lane ? ", slab layout: " + std::to_string(bank_cost) + " conflict cycles per step in the model" : std::string{})
                + (ev_inline ? "; event equations, final step size and state update inside the stepper" : "");
    return ret;
}

} // namespace

emitted_module emit_cluster_v2(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    // The one-lane pair kernel first (pair-pattern systems without runtime parameters, not the stepper with events); if
    // its shape requirements fail or the jets of its systems do not fit in LDS, the lane-pair / pipelined kernels.
    // Few pairs (3 or 6: model::nbody(3), (4)): with 4 / 8 lanes per system a CU holds 128 / 64 systems, whose jets do not
    // fit in its LDS - and the lane-pair kernel with the jets in global scratch which used to serve them spills 176 / 116
    // registers. More lanes per system than pairs (idle lanes replicate pair 0 and write to dummy slots, like the 16th lane
    // of the outer Solar System) bring the systems per CU down to what fits: the second and third attempts.
    bool in_lds = false;
    std::string why1;
    const bool try_one_lane = opts.cluster_kernel == 0 || opts.cluster_kernel == 5;
    for (const std::uint32_t min_lanes : {4u, 8u, 16u}) {
        why1.clear();
        auto ret = emit_cluster_v2_impl(p, opts, why1, try_one_lane, in_lds, min_lanes);
        const bool is_v5 = ret.notes.find("cluster mode v5") != std::string::npos;
        if (try_one_lane) {
            detail::log_message(log_level::debug, "one-lane-per-pair kernel, at least " + std::to_string(min_lanes) + " lanes per system: "
                                              + (!why1.empty() ? why1 : (!is_v5 ? "not applicable" : (in_lds ? "accepted" : "the jets of the systems of a CU do not fit in its LDS"))));
        }
        if (why1.empty() && (in_lds || !is_v5)) {
            why_not.clear();
            return ret;
        }
        // (Another attempt only for the reasons which more lanes per system cure.)
        const bool lds_reason = why1.find("needs the jets in LDS") != std::string::npos;
        if (!try_one_lane || (!why1.empty() && !lds_reason) || (why1.empty() && !is_v5)) {
            break;
        }
    }
    return emit_cluster_v2_impl(p, opts, why_not, false, in_lds);
}

} // namespace heyoka_amd
