// HIP code generation, "block" mode: one ODE system per workgroup.
//
// Target: decompositions with many isomorphic nonlinear clusters, e.g. model::nbody(64) (2016 body pairs,
// 18 663 u variables; BASELINE config 5), where a system cannot live in one wavefront: the jets of the
// clusters (2016 x 5 x 20 doubles = 1.6 MB) exceed the register file *and* the LDS of a compute unit.
// With one lane per system (table mode) the tape is 3 MB per lane, the ensemble exposes only one
// wavefront per SIMD and every convolution term is an uncoalesced-by-design HBM access.
//
// Here the 256 lanes of a workgroup cooperate on one system at a time:
//  * cluster phase (order k): lane = cluster (strided over the clusters). The lane loads the lower-order
//    coefficients of the cluster's "stored" members from a per-workgroup tape in global memory, laid out
//    tape[(member * order + m) * n_clusters + cluster] so that the lanes of a wave read consecutive
//    addresses (fully coalesced 512 B requests; the tape of the systems in flight is L2 / MALL sized),
//    evaluates the cluster's recurrences with the same node emitters as the other modes, stores the new
//    coefficients and publishes the cluster outputs to an LDS slab;
//  * glue phases: lane = glue node (sums / differences / scalings of current-order values), grouped by
//    dependency level and shape, operands and results in the LDS slab, slots from tables in global memory
//    (coalesced, L2-resident, shared by all the workgroups);
//  * state-variable recursion x^[k+1] = rhs^[k] / (k + 1), lane = state variable; the jets of the state
//    variables go to a small per-workgroup array in global memory for the final Horner / compensated
//    update, again with lane = state variable;
//  * infinity norms for the step-size selector: wave shuffles + one LDS exchange.
// Workgroups are persistent and pull systems from the device-side work queue.
//
// Reference semantics of every phase as in the other modes:
// taylor_compute_jet (src/taylor_02.cpp:1339-1418), taylor_determine_h (src/taylor_00.cpp:102-273),
// taylor_run_multihorner / taylor_run_ceval (:279-460), step / propagate semantics
// (src/taylor_adaptive_batch.cpp:632-727, :1137-1534).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <sstream>
#include <set>
#include <numeric>
#include <string>
#include <vector>

#include "hip_emit_cluster_plan.hpp"
#include "hip_emit_detail.hpp"
#include "logging.hpp"

namespace heyoka_amd
{

using namespace cluster_detail;
using emit_detail::ssa_emitter;

emitted_module emit_block(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    using emit_detail::prelude;
    using emit_detail::rhofac;

    emitted_module ret;
    cluster_plan pl;
    plan_limits lim;
    lim.max_clusters = 1u << 20;
    lim.jets_in_registers = false;
    lim.absorb_linear = !opts.block_no_absorb;
    why_not = make_plan(p, opts.order, pl, lim);
    if (!why_not.empty()) {
        return ret;
    }

    const auto n_eq = p.n_eq, order = opts.order;
    // Slots of the exported cluster members: output-major blocks of n_clusters slots each. When the number of clusters is a
    // multiple of the 32 bank pairs of LDS (2016 for nbody(64)), the three products of a pair sit in ONE bank, and so do
    // the operands of the sums of the three coordinates of a body, which the glue gathers on adjacent lanes: one slot of
    // padding between the blocks (an odd distance). ("nopad": A/B harness.)
    if (pl.classes.size() <= 1u && pl.clusters.size() % 2u == 0u && pl.out_pos.size() > 1u
        && ("," + opts.dev.block_opts + ",").find(",nopad,") == std::string::npos) {
        const auto ncl = static_cast<int>(pl.clusters.size()), nout = static_cast<int>(pl.out_pos.size());
        const auto first = static_cast<int>(n_eq), last = first + ncl * nout;
        bool layout_ok = true;
        for (int c = 0; c < ncl && layout_ok; ++c) {
            for (int q = 0; q < nout; ++q) {
                layout_ok = layout_ok && pl.slot_of[pl.clusters[static_cast<std::size_t>(c)][pl.out_pos[static_cast<std::size_t>(q)]]] == first + q * ncl + c;
            }
        }
        if (layout_ok) {
            for (auto &sl : pl.slot_of) {
                if (sl >= last) {
                    sl += nout - 1;
                } else if (sl >= first) {
                    sl += (sl - first) / ncl;
                }
            }
            pl.n_slots += static_cast<std::uint32_t>(nout - 1);
        }
    }
    // Workgroup size: 256 lanes = one wavefront per SIMD with the whole register file (512 VGPRs) for the
    // history of the cluster being processed.
    std::uint32_t bs = 256;
    {
        // Small cluster counts: do not park lanes (several smaller workgroups fit on a compute unit).
        const auto nc0 = static_cast<std::uint32_t>(pl.clusters.size());
        if (nc0 <= 128u) {
            bs = 128;
        }
        // (HEYOKA_AMD_BLOCK_OPTS=bs=N: A/B harness.)
        std::string s = "," + opts.dev.block_opts + ",";
        std::replace(s.begin(), s.end(), ':', ',');
        if (const auto pos = s.find(",bs="); pos != std::string::npos) {
            bs = static_cast<std::uint32_t>(std::atoi(s.c_str() + pos + 4u));
        }
    }
    const auto nc = static_cast<std::uint32_t>(pl.clusters.size());
    const auto ncp = (nc + 63u) / 64u * 64u;
    const auto &t0 = pl.clusters[0];
    const auto n_ext = static_cast<std::uint32_t>(pl.ext_u[0].size());
    const auto n_out = static_cast<std::uint32_t>(pl.out_pos.size());
    const auto n_cst = static_cast<std::uint32_t>(pl.cst_pos.size());
    const auto n_sto = static_cast<std::uint32_t>(pl.stored_pos.size());
    const auto n_slots = pl.n_slots;

    // Registers: the history of the stored members is loaded at every order.
    if (static_cast<std::uint64_t>(n_sto) * order * 2u > 440u) {
        why_not = "the jets of a cluster do not fit in the register file";
        return ret;
    }
    // Stored members which are linear functions of external inputs only (e.g. the coordinate differences of
    // an N-body pair) are not written to the tape: their lower-order coefficients are recomputed from the
    // jets of the external inputs, which are kept in LDS (ejet[m * n_ej + e]). This removes 3 of the 5 tape
    // streams of the N-body clusters.
    std::vector<char> recomp(n_sto, 0);
    std::vector<char> ext_used(n_ext, 0);
    std::vector<std::uint32_t> ej_slots; // distinct slab slots of the external inputs with LDS jets
    std::vector<std::uint32_t> ej_index; // [x * nc + c] -> index into ej_slots
    {
        std::map<std::uint32_t, std::uint32_t> ext_pos;
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            ext_pos[pl.ext_u[0][x]] = x;
        }
        for (std::uint32_t s = 0; s < n_sto; ++s) {
            const auto &n = p.nodes[t0[pl.stored_pos[s]] - n_eq];
            bool ok = (n.kind == func_kind::sub || n.kind == func_kind::sum
                       || (n.kind == func_kind::prod && n.args.size() == 2u && !is_var(n.args[0])));
            bool any_var = false;
            for (const auto &o : n.args) {
                if (is_var(o)) {
                    any_var = true;
                    ok = ok && ext_pos.count(o.idx) != 0u;
                }
            }
            if (ok && any_var) {
                recomp[s] = 1;
                for (const auto &o : n.args) {
                    if (is_var(o)) {
                        ext_used[ext_pos[o.idx]] = 1;
                    }
                }
            }
        }
        // (Jets in the order of the external inputs of the template, within one input by ascending slot: the layout does not
        // depend on the order of the clusters.)
        std::map<std::uint32_t, std::uint32_t> slot_to_ej;
        ej_index.assign(static_cast<std::size_t>(n_ext) * nc, 0u);
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            if (ext_used[x] == 0) {
                continue;
            }
            std::set<std::uint32_t> fresh;
            for (std::uint32_t c = 0; c < nc; ++c) {
                const auto slot = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][x]]);
                if (slot_to_ej.count(slot) == 0u) {
                    fresh.insert(slot);
                }
            }
            for (const auto slot : fresh) {
                slot_to_ej.emplace(slot, static_cast<std::uint32_t>(ej_slots.size()));
                ej_slots.push_back(slot);
            }
            for (std::uint32_t c = 0; c < nc; ++c) {
                ej_index[static_cast<std::size_t>(x) * nc + c] = slot_to_ej[static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][x]])];
            }
        }
        // Order of the clusters (round 6): lane l of a wavefront reads the jets of the inputs of ITS cluster from LDS - 64-bit
        // reads, served 32 lanes at a time, one cycle when the addresses of a group of 32 lanes fall into distinct banks
        // (or coincide). In the order of the decomposition (pairs of an N-body system: body-major) a group of 32 straddles
        // the boundary between two bodies and its partners collide: 1.24 cycles per group on nbody(64), a quarter of the
        // LDS cycles of the kernel. HEYOKA_AMD_BLOCK_OPTS=perm deals the clusters greedily into groups of 32 whose inputs are
        // conflict-free input by input (1.01 on nbody(64)) - MEASURED: 1.506e6 against 1.525e6 system-steps/s in the order
        // of the decomposition (profiles/r06_nbody64_experiments.log): the conflicts of these reads are not what binds;
        // the order of the decomposition stays the default.
        if (!ej_slots.empty() && nc >= 64u && ("," + opts.dev.block_opts + ",").find(",perm,") != std::string::npos) {
            std::vector<std::uint32_t> rem(nc), order_;
            std::iota(rem.begin(), rem.end(), 0u);
            order_.reserve(nc);
            while (!rem.empty()) {
                std::vector<std::uint32_t> grp, keep;
                std::vector<std::map<std::uint32_t, std::uint32_t>> bank(n_ext);
                for (const auto c : rem) {
                    bool ok = grp.size() < 32u;
                    for (std::uint32_t x = 0; x < n_ext && ok; ++x) {
                        if (ext_used[x] != 0) {
                            const auto a = ej_index[static_cast<std::size_t>(x) * nc + c];
                            const auto it = bank[x].find(a % 32u);
                            ok = it == bank[x].end() || it->second == a;
                        }
                    }
                    if (ok) {
                        grp.push_back(c);
                        for (std::uint32_t x = 0; x < n_ext; ++x) {
                            if (ext_used[x] != 0) {
                                const auto a = ej_index[static_cast<std::size_t>(x) * nc + c];
                                bank[x][a % 32u] = a;
                            }
                        }
                    } else {
                        keep.push_back(c);
                    }
                }
                std::size_t taken = 0;
                while (grp.size() < 32u && taken < keep.size()) {
                    grp.push_back(keep[taken++]);
                }
                keep.erase(keep.begin(), keep.begin() + static_cast<std::ptrdiff_t>(taken));
                order_.insert(order_.end(), grp.begin(), grp.end());
                rem.swap(keep);
            }
            // new position -> old cluster: apply to everything indexed by the cluster.
            std::vector<std::uint32_t> new_of(nc);
            for (std::uint32_t i = 0; i < nc; ++i) {
                new_of[order_[i]] = i;
            }
            const auto permute = [&](auto &v) {
                if (v.size() == nc) {
                    // (Element by element: references to the elements - the template cluster - stay valid.)
                    const auto w = v;
                    for (std::uint32_t i = 0; i < nc; ++i) {
                        v[i] = w[order_[i]];
                    }
                }
            };
            permute(pl.clusters);
            permute(pl.ext_u);
            permute(pl.cst_val);
            permute(pl.par_idx);
            for (auto &c : pl.cluster_of) {
                if (c >= 0) {
                    c = static_cast<int>(new_of[static_cast<std::uint32_t>(c)]);
                }
            }
            for (auto &cls : pl.classes) {
                for (auto &m : cls.members) {
                    m = new_of[m];
                }
            }
            auto ej2 = ej_index;
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                for (std::uint32_t i = 0; i < nc; ++i) {
                    ej2[static_cast<std::size_t>(x) * nc + i] = ej_index[static_cast<std::size_t>(x) * nc + order_[i]];
                }
            }
            ej_index.swap(ej2);
        }
    }
    auto n_ej = static_cast<std::uint32_t>(ej_slots.size());
    const auto lds_for = [&](std::uint32_t nej) {
        return static_cast<std::uint64_t>(n_slots) * 8u + static_cast<std::uint64_t>(nej) * (order - 1u) * 8u + 128u;
    };
    if (lds_for(n_ej) > 150u * 1024u) {
        // No room for the jets of the external inputs: everything goes through the tape.
        std::fill(recomp.begin(), recomp.end(), 0);
        std::fill(ext_used.begin(), ext_used.end(), 0);
        ej_slots.clear();
        n_ej = 0;
    }
    if (lds_for(n_ej) > 150u * 1024u) {
        why_not = "the exchange slab does not fit in LDS";
        return ret;
    }
    // Tape rows: only the members which are not recomputed.
    std::vector<std::uint32_t> tape_row(n_sto, 0u);
    std::uint32_t n_tape = 0;
    for (std::uint32_t s = 0; s < n_sto; ++s) {
        if (recomp[s] == 0) {
            tape_row[s] = n_tape++;
        }
    }
    for (const auto &d : p.sv_defs) {
        if (d.type == operand::kind::uvar && pl.slot_of[d.idx] < 0) {
            why_not = "state variable definition without a slot";
            return ret;
        }
    }

    // ---- tables ----
    const bool wide = n_slots > 65535u;
    const std::string ut = wide ? "unsigned" : "unsigned short";
    std::ostringstream tbl;
    std::vector<std::pair<std::string, std::size_t>> utbl_list; // (name, entries) of the index tables
    const auto emit_utbl = [&](const std::string &name, const std::vector<std::uint32_t> &v) {
        utbl_list.emplace_back(name, std::max<std::size_t>(v.size(), 1u));
        tbl << "__device__ const " << ut << " __attribute__((aligned(16))) " << name << "[" << std::max<std::size_t>(v.size(), 1u) << "] = {";
        for (const auto x : v) {
            tbl << x << ",";
        }
        tbl << "};\n";
    };
    const auto emit_dtbl = [&](const std::string &name, const std::vector<double> &v) {
        tbl << "__device__ const double " << name << "[" << std::max<std::size_t>(v.size(), 1u) << "] = {";
        for (const auto x : v) {
            tbl << fp_literal(x) << ",";
        }
        tbl << "};\n";
    };
    {
        std::vector<std::uint32_t> v(static_cast<std::size_t>(n_ext) * nc);
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            for (std::uint32_t c = 0; c < nc; ++c) {
                v[static_cast<std::size_t>(x) * nc + c] = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][x]]);
            }
        }
        emit_utbl("hy_ext", v);
        v.assign(static_cast<std::size_t>(n_out) * nc, 0u);
        for (std::uint32_t x = 0; x < n_out; ++x) {
            for (std::uint32_t c = 0; c < nc; ++c) {
                v[static_cast<std::size_t>(x) * nc + c]
                    = static_cast<std::uint32_t>(pl.slot_of[pl.clusters[c][pl.out_pos[x]]]);
            }
        }
        emit_utbl("hy_out", v);
        std::vector<double> dv(static_cast<std::size_t>(n_cst) * nc);
        for (std::uint32_t x = 0; x < n_cst; ++x) {
            for (std::uint32_t c = 0; c < nc; ++c) {
                dv[static_cast<std::size_t>(x) * nc + c] = pl.cst_val[c][x];
            }
        }
        emit_dtbl("hy_cst", dv);
        emit_utbl("hy_extj", n_ej != 0u ? ej_index : std::vector<std::uint32_t>{});
        emit_utbl("hy_ejs", ej_slots);
    }
    // Glue groups: per argument a table of slots (variables) or of values (non-structural numbers).
    for (std::size_t g = 0; g < pl.groups.size(); ++g) {
        const auto &grp = pl.groups[g];
        const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
        for (std::size_t a = 0; a < n0.args.size(); ++a) {
            const auto name = "hy_g" + std::to_string(g) + "_a" + std::to_string(a);
            if (is_var(n0.args[a])) {
                std::vector<std::uint32_t> v;
                for (const auto u : grp.nodes) {
                    v.push_back(static_cast<std::uint32_t>(pl.slot_of[p.nodes[u - n_eq].args[a].idx]));
                }
                emit_utbl(name, v);
            } else if (n0.args[a].type == operand::kind::num) {
                std::vector<double> v;
                for (const auto u : grp.nodes) {
                    v.push_back(p.nodes[u - n_eq].args[a].value);
                }
                emit_dtbl(name, v);
            }
        }
        std::vector<std::uint32_t> v;
        for (const auto u : grp.nodes) {
            v.push_back(static_cast<std::uint32_t>(pl.slot_of[u]));
        }
        emit_utbl("hy_g" + std::to_string(g) + "_o", v);
    }
    // State-variable definitions: kind 0 = u variable (slot), 1 = number, 2 = parameter.
    {
        std::vector<std::uint32_t> kind(n_eq), idx(n_eq);
        std::vector<double> val(n_eq, 0.);
        for (std::uint32_t i = 0; i < n_eq; ++i) {
            const auto &d = p.sv_defs[i];
            if (d.type == operand::kind::uvar) {
                kind[i] = 0;
                idx[i] = static_cast<std::uint32_t>(pl.slot_of[d.idx]);
            } else if (d.type == operand::kind::num) {
                kind[i] = 1;
                val[i] = d.value;
            } else {
                kind[i] = 2;
                idx[i] = d.idx;
            }
        }
        emit_utbl("hy_sv_kind", kind);
        emit_utbl("hy_sv_idx", idx);
        emit_dtbl("hy_sv_val", val);
    }

    // ---- "v2" cluster phase (point-mass pair clusters): see emit_pair_rounds() below ----
    // The order loop is ROLLED and the rounds of a lane (its n_iter clusters) are unrolled, so that the coefficients of
    // order < M of the two members with a recurrence on themselves (r^2 and its power) stay in REGISTERS for the whole
    // step (2 * M doubles per cluster of the lane); the convolutions are evaluated as index pairs (i, k - i), i < k - i:
    // the low index from the registers (or, for M <= i, from the tape), the high index from the tape - every tape row
    // >= M is read once per order instead of every row, and the rows just written are the ones read next (L2).
    pair_pattern pp;
    detect_pair_pattern(p, pl, pp);
    const std::uint32_t v2_T = (order - 1u) / 2u + 1u; // index pairs (slots) of the highest order
    std::uint32_t v2_M = 0, v2_hand = 1;
    bool v2 = [&]() {
        if (!opts.dev.block_v2) {
            return false;
        }
        if (!pp.ok || pp.sc != -1 || pp.rx[0] != -1 || pp.rx[1] != -1 || pp.rx[2] != -1 || n_cst != 0u
            || !pl.par_pos.empty() || pl.glue_has_par || wide || pl.cluster_level != 1u || n_ej == 0u || n_tape != 2u
            || n_out != 3u || order < 6u || p.n_par != 0u) {
            detail::log_message(log_level::debug,
                                "block mode, v2 cluster phase not applicable: pair pattern " + std::to_string(pp.ok) + ", scaling "
                                    + std::to_string(pp.sc) + ", reactions " + std::to_string(pp.rx[0]) + ", constants "
                                    + std::to_string(n_cst) + ", wide " + std::to_string(wide) + ", cluster level "
                                    + std::to_string(pl.cluster_level) + ", external jets " + std::to_string(n_ej) + ", tape members "
                                    + std::to_string(n_tape) + ", outputs " + std::to_string(n_out));
            return false;
        }
        if (static_cast<std::uint64_t>(n_ej) * order * 8u >= 65536u || (nc + bs - 1u) / bs > 16u) {
            return false;
        }
        for (const auto c : constant_uvars(p)) {
            if (c != 0) {
                return false;
            }
        }
        // External inputs: state variables (their order-k values are recorded by the state recursion).
        for (std::uint32_t c = 0; c < nc; ++c) {
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                const auto u = pl.ext_u[c][x];
                if (u >= n_eq || pl.slot_of[u] != static_cast<int>(u)) {
                    return false;
                }
            }
        }
        // Glue: nodes whose order-k rule is the same for every k >= 1.
        for (const auto &grp : pl.groups) {
            const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
            bool ok = false;
            if (n0.kind == func_kind::sum || n0.kind == func_kind::sub) {
                ok = true;
            } else if (n0.kind == func_kind::prod && n0.args.size() == 2u && n0.args[0].type == operand::kind::num
                       && is_var(n0.args[1])) {
                ok = true;
            }
            if (!ok) {
                return false;
            }
        }
        return true;
    }();
    // Two wavefronts per SIMD for the v2 cluster phase with more than four rounds per lane (round 6): 512 lanes with 256
    // registers each. With eight rounds on one wavefront per SIMD the kernel was bound by the ISSUE of its single
    // wavefront (8.5 cycles per instruction by the counters: 30 % VALU, 13 % LDS, 5 % memory, the rest waiting); HBM bytes,
    // LDS reads and tape loads of the slots each removed in turn gained 11 ... 13 % only (profiles/r06_nbody64_experiments.log).
    // With a second wavefront the ping-pong of the LDS operands is not needed (the other wavefront covers the latency; its
    // registers go to the rows in registers), groups of two rounds, tape loads two slots ahead: nbody(64) 1.42e6 -> 1.52e6.
    bool v2_two_waves = false, v2_recip = false, v2_fuse_last = false, v2_dbuf = false, v2_lean = false;
    if (v2 && bs == 256u && nc > 1024u && opts.dev.block_opts.find("bs=") == std::string::npos) {
        bs = 512;
        v2_two_waves = true;
    }
    // ---- body ----
    ssa_emitter e(p, order);
    auto &os = e.os;
    for (std::uint32_t x = 0; x < n_cst; ++x) {
        const auto [q, a] = pl.cst_pos[x];
        e.numpar_override[&p.nodes[t0[q] - n_eq].args[a]] = "ccst" + std::to_string(x);
    }
    const auto sync = [&]() { os << "__syncthreads();\n"; };

    // Cluster phase at order k: a rolled loop over the clusters of the lane. The loads of a cluster (slot
    // indices, per-cluster constants, tape history) can be software-pipelined, i.e. requested one round ahead:
    //   pf_mode 0 - no pipelining: everything is loaded at the top of the round (default);
    //   pf_mode 1 - only the slot indices / constants are requested one round ahead;
    //   pf_mode 2 - the tape history as well.
    // Measured on an MI355X (nbody64, 65 536 systems): mode 0 8.4e5, mode 2 5.0e5 system-steps/s - the
    // loop-carried copies of the history push the kernel over the 512-VGPR budget and the spill traffic
    // costs more than the latency it hides.
    const auto n_iter = (nc + bs - 1u) / bs;
    const int pf_mode = [&]() {
        if (n_iter <= 1u) {
            return 0;
        }
        return 0;
    }();
    const std::uint32_t sched_every = 4;
    const auto emit_cluster = [&](std::uint32_t k) {
        if (n_ej != 0u && k + 1u < order) {
            // Record the order-k coefficients of the external inputs (read back from order k + 1 on).
            os << "for (unsigned e = tid; e < " << n_ej << "u; e += " << bs << "u) ejet["
               << static_cast<std::uint64_t>(k) * n_ej << "u + e] = slab[hy_ejs[e]];\n";
        }
        // Names of the per-cluster loads: (variable name, type, load expression in terms of `cc`).
        struct pf {
            std::string name, type, expr;
            bool carried = false; // requested one round ahead
        };
        std::vector<pf> loads;
        for (std::uint32_t x = 0; x < n_cst; ++x) {
            loads.push_back({"ccst" + std::to_string(x), "double",
                             "hy_cst[" + std::to_string(static_cast<std::uint64_t>(x) * nc) + "u + cc]"});
        }
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            loads.push_back({"ie" + std::to_string(x), "unsigned",
                             "hy_ext[" + std::to_string(static_cast<std::uint64_t>(x) * nc) + "u + cc]"});
        }
        for (std::uint32_t x = 0; x < n_out; ++x) {
            loads.push_back({"io" + std::to_string(x), "unsigned",
                             "hy_out[" + std::to_string(static_cast<std::uint64_t>(x) * nc) + "u + cc]"});
        }
        if (n_ej != 0u && k > 0u) {
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                if (ext_used[x] != 0) {
                    loads.push_back({"ej" + std::to_string(x), "unsigned",
                                     "hy_extj[" + std::to_string(static_cast<std::uint64_t>(x) * nc) + "u + cc]"});
                }
            }
        }
        for (std::uint32_t s = 0; s < n_sto; ++s) {
            if (recomp[s] != 0) {
                continue;
            }
            const auto u = t0[pl.stored_pos[s]];
            for (std::uint32_t m = 0; m < k; ++m) {
                const auto nm = "h" + std::to_string(s) + "_" + std::to_string(m);
                loads.push_back({nm, "double",
                                 "tape[" + std::to_string((static_cast<std::uint64_t>(tape_row[s]) * order + m) * ncp)
                                     + "u + cc]",
                                 pf_mode == 2});
                e.val(u, m) = nm;
            }
        }
        for (auto &l : loads) {
            if (l.type != "double" || l.name.rfind("ccst", 0) == 0) {
                l.carried = pf_mode >= 1;
            }
        }
        // Prologue: carried loads of the first cluster of the lane (clamped: the idle lanes of a partial last
        // round replicate the last cluster and do not write anything).
        os << "{\n";
        os << "unsigned c = tid;\n";
        for (const auto &l : loads) {
            if (l.carried) {
                os << l.type << " " << l.name << ";\n";
            }
        }
        os << "{\nconst unsigned cc = c < " << nc << "u ? c : " << nc - 1u << "u;\n";
        for (const auto &l : loads) {
            if (l.carried) {
                os << l.name << " = " << l.expr << ";\n";
            }
        }
        os << "}\n";
        os << "#pragma nounroll\n";
        os << "for (unsigned it = 0; it < " << n_iter << "u; ++it, c += " << bs << "u) {\n";
        os << "const bool live = c < " << nc << "u;\n";
        os << "const unsigned cw = live ? c : " << nc - 1u << "u;\n";
        for (const auto &l : loads) {
            if (!l.carried) {
                auto ex = l.expr;
                const auto pos = ex.rfind("cc]");
                ex.replace(pos, 2, "cw");
                os << "const " << l.type << " " << l.name << " = " << ex << ";\n";
            }
        }
        if (pf_mode != 0) {
            os << "const unsigned cc = (c + " << bs << "u) < " << nc << "u ? (c + " << bs << "u) : " << nc - 1u
               << "u;\n";
            for (const auto &l : loads) {
                if (l.carried) {
                    os << "const " << l.type << " n_" << l.name << " = " << l.expr << ";\n";
                }
            }
        }
        // Lower-order coefficients recomputed from the LDS jets of the external inputs.
        if (n_ej != 0u && k > 0u) {
            for (std::uint32_t m = 0; m < k; ++m) {
                for (std::uint32_t x = 0; x < n_ext; ++x) {
                    if (ext_used[x] != 0) {
                        e.val(pl.ext_u[0][x], m)
                            = e.def("ejet[" + std::to_string(static_cast<std::uint64_t>(m) * n_ej) + "u + ej"
                                    + std::to_string(x) + "]");
                    }
                }
                for (std::uint32_t s = 0; s < n_sto; ++s) {
                    if (recomp[s] != 0) {
                        e.node(t0[pl.stored_pos[s]] - n_eq, m);
                    }
                }
                // NOTE: without a fence the scheduler hoists all the LDS reads of the inputs' jets to the top
                // (2 * 3 * k live doubles on top of the history itself): hundreds of spilled registers.
                if (m % sched_every == sched_every - 1u) {
                    os << "__builtin_amdgcn_sched_barrier(0);\n";
                }
            }
            os << "__builtin_amdgcn_sched_barrier(0);\n";
        }
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            e.val(pl.ext_u[0][x], k) = e.def("slab[ie" + std::to_string(x) + "]");
        }
        for (const auto u : t0) {
            e.node(u - n_eq, k);
        }
        os << "if (live) {\n";
        if (k + 1u < order) {
            for (std::uint32_t s = 0; s < n_sto; ++s) {
                if (recomp[s] == 0) {
                    os << "tape[" << (static_cast<std::uint64_t>(tape_row[s]) * order + k) * ncp
                       << "u + cw] = " << e.val(t0[pl.stored_pos[s]], k) << ";\n";
                }
            }
        }
        for (std::uint32_t x = 0; x < n_out; ++x) {
            os << "slab[io" << x << "] = " << e.val(t0[pl.out_pos[x]], k) << ";\n";
        }
        os << "}\n";
        for (const auto &l : loads) {
            if (l.carried) {
                os << l.name << " = n_" << l.name << ";\n";
            }
        }
        os << "}\n}\n";
    };

    // Glue group at order k: branch-free rounds (clamped node index, the result of an idle lane goes to a dummy
    // slot), so that all the groups of a level form one basic block and their table / LDS loads overlap.
    const auto emit_glue_group = [&](std::size_t g, std::uint32_t k) {
        const auto &grp = pl.groups[g];
        const auto rep = grp.nodes[0];
        const auto &n0 = p.nodes[rep - n_eq];
        const auto gn = "hy_g" + std::to_string(g);
        const auto ng = static_cast<std::uint32_t>(grp.nodes.size());
        for (std::uint32_t r = 0; r * bs < ng; ++r) {
            os << "{\nconst unsigned jr = tid + " << r * bs << "u;\n";
            const bool partial = (r + 1u) * bs > ng;
            if (partial) {
                os << "const bool ok = jr < " << ng << "u;\nconst unsigned j = ok ? jr : " << ng - 1u << "u;\n";
            } else {
                os << "const unsigned j = jr;\n";
            }
            const auto saved = e.numpar_override;
            std::vector<std::pair<std::uint32_t, std::string>> saved_vals;
            for (std::size_t a = 0; a < n0.args.size(); ++a) {
                const auto &o = n0.args[a];
                if (is_var(o)) {
                    const auto nm = e.def("slab[" + gn + "_a" + std::to_string(a) + "[j]]");
                    saved_vals.emplace_back(o.idx, e.val(o.idx, k));
                    e.val(o.idx, k) = nm;
                } else if (o.type == operand::kind::num) {
                    e.numpar_override[&o] = gn + "_a" + std::to_string(a) + "[j]";
                }
            }
            if (n0.kind == func_kind::prod && n0.args[0].type == operand::kind::num && n0.args[0].value == -1.) {
                e.numpar_override.erase(&n0.args[0]);
            }
            e.node(rep - n_eq, k);
            if (partial) {
                os << "slab[ok ? (unsigned)" << gn << "_o[j] : " << n_slots << "u] = " << e.val(rep, k) << ";\n";
            } else {
                os << "slab[" << gn << "_o[j]] = " << e.val(rep, k) << ";\n";
            }
            for (auto it = saved_vals.rbegin(); it != saved_vals.rend(); ++it) {
                e.val(it->first, k) = it->second;
            }
            e.numpar_override = saved;
            os << "}\n";
        }
    };

    const auto sv_rounds = (n_eq + bs - 1u) / bs;
    std::string v2_decl; // kernel-lifetime declarations of the v2 cluster phase

    // The whole step body of the v2 cluster phase (see the eligibility test above). Index pairs ("slots"): at order k
    // the convolutions of the cluster run over the pairs (i, p = k - i), 0 <= i <= p. Slot 0 pairs the order-0
    // coefficients with the order-k ones (which are being computed), slots 1 .. floor(k / 2) pair known coefficients;
    // for an even k the last slot is the middle term (i = p), entered with the weights (1/2, 0) on the symmetric parts.
    //   r2^[k]  = 2 * sum_i d^[i] . d^[p]                                                   (src/detail/sum_sq.cpp:100-245)
    //   pw^[k]  = (sum_j (k a - j (a + 1)) r2^[k-j] pw^[j]) / (k r2^[0])                   (src/math/pow.cpp:395-550)
    //   f_c^[k] = sum_j d_c^[k-j] pw^[j]                                                    (src/math/prod.cpp:314-395)
    // Data of a slot: r2^[i], pw^[i] from the registers (i < M) or the tape, r2^[p], pw^[p] from the tape, d^[i], d^[p]
    // recomputed from the jets of the external inputs in LDS (rows stored in REVERSE order, so that the row of p = k - i
    // is a non-negative constant offset from the row of k). All the tape loads of a round are issued, unconditionally
    // (rows of slots beyond the last one are clamped to a row which is loaded anyway), one round ahead of their use.
    const auto emit_pair_rounds_body = [&]() -> bool {
        const auto R = n_iter;
        const auto T = v2_T;
        // Developer experiments (timing only, wrong results): 1 = no slots beyond slot 0, 2 = no glue arithmetic, 3 = no LDS
        // reads in the slots, 4 = no tape loads in the slots.
        // HEYOKA_AMD_BLOCK_OPTS (A/B harness): "exp=N" (the experiments above), "M=N" (rows in registers), "tape_low" (round-5
        // behaviour: the high member of a slot always comes from the tape, every row is stored).
        const auto bopt = [&](const std::string &name, int dflt) {
            std::string s = "," + opts.dev.block_opts + ",";
            std::replace(s.begin(), s.end(), ':', ',');
            if (const auto pos = s.find("," + name + "="); pos != std::string::npos) {
                return std::atoi(s.c_str() + pos + name.size() + 2u);
            }
            return s.find("," + name + ",") != std::string::npos ? 1 : dflt;
        };
        const int exp_mode = bopt("exp", 0);
        std::uint32_t M = 160u / (4u * R);
        // Two wavefronts per SIMD: 96 registers for rows which stay in registers - the low rows 0 ... M - 1 for the whole step
        // and the `hand` most recent ones, handed over from order to order (below). A low row m saves 20 - 2 m tape reads per
        // step, a handed-over row h 21 - 2 h: dealt in turn (row 0, m = 1, h = 2, m = 2, h = 3, ...).
        std::uint32_t hand_dflt = 1;
        // (More than four rounds per lane - nbody(80): seven -: per-round registers cost too much, measured: spills inside
        // the order loop halve the rate; two rows, one hand-over, no differences / descriptors kept.)
        v2_lean = v2_two_waves && R > 4u;
        if (v2_lean) {
            M = std::max(2u, 80u / (4u * R));
        } else if (v2_two_waves) {
            const auto rows = std::max(3u, 96u / (4u * R));
            M = 1;
            for (std::uint32_t e = 0; e + 1u < rows; ++e) {
                if (e % 2u == 0u) {
                    ++M;
                } else {
                    ++hand_dflt;
                }
            }
        }
        M = static_cast<std::uint32_t>(bopt("M", static_cast<int>(M)));
        M = std::min(T, std::max(2u, M));
        v2_M = M;
        // The high member p = k - i of slot i is one of the rows in registers while k < i + M (slots 2 ... M - 1 of the orders
        // 2 i ... i + M - 1): selected from the register copies by the (wave-uniform) order, not loaded. The rows < M are
        // then never read from the tape and not stored; neither is the row of order P - 2, which only order P - 1 reads -
        // from the hand-over registers. 11 of the 120 row transfers of a step of nbody(64) (rows < 5 in registers).
        const bool reg_high = bopt("reg_high", 0) != 0 && exp_mode == 0;
        // (reg_high=2: the selection behind a wave-uniform branch at the head of the slot - no instruction at the orders
        // whose high member comes from the tape.)
        const bool reg_high_br = bopt("reg_high", 0) == 2;
        // Quotients as products with reciprocals + one exact-residual correction unless kw::exact_division ("div": A/B).
        const bool recip = !opts.exact_division && bopt("div", 0) == 0;
        v2_recip = recip;
        int s_sq = -1, s_pw = -1;
        for (std::uint32_t s = 0; s < n_sto; ++s) {
            if (recomp[s] == 0 && pl.stored_pos[s] == pp.sq) {
                s_sq = static_cast<int>(s);
            }
            if (recomp[s] == 0 && pl.stored_pos[s] == pp.pw) {
                s_pw = static_cast<int>(s);
            }
        }
        int io_of[3] = {-1, -1, -1};
        for (std::uint32_t c = 0; c < 3u; ++c) {
            for (std::uint32_t x = 0; x < n_out; ++x) {
                if (pl.out_pos[x] == pp.pr[c]) {
                    io_of[c] = static_cast<int>(x);
                }
            }
            for (std::uint32_t s = 0; s < n_sto; ++s) {
                if (pl.stored_pos[s] == pp.d[c] && recomp[s] == 0) {
                    return false;
                }
            }
        }
        if (s_sq < 0 || s_pw < 0 || io_of[0] < 0 || io_of[1] < 0 || io_of[2] < 0 || n_ext > 8u) {
            return false;
        }
        const auto arow0 = static_cast<std::uint64_t>(tape_row[static_cast<std::uint32_t>(s_sq)]) * order;
        const auto brow0 = static_cast<std::uint64_t>(tape_row[static_cast<std::uint32_t>(s_pw)]) * order;
        // ("split": A/B harness - the two tape members in separate halves of the tape, two 8-byte accesses.)
        const bool il_tape = bopt("split", 0) == 0 && bopt("xpf", 0) == 0 && n_tape == 2u;
        const auto rowb = static_cast<std::uint64_t>(ncp) * 8u;
        const auto P = order;
        const auto ejrow = static_cast<std::uint64_t>(n_ej) * 8u;         // bytes per row of the input jets
        const auto ej0b = static_cast<std::uint64_t>(P - 1u) * ejrow;     // row of order 0
        const double ex = pp.ex, e1 = pp.ex + 1.;
        const auto U = [](std::uint64_t x) { return std::to_string(x) + "u"; };
        const auto S = [](std::uint32_t x) { return std::to_string(x); };
        const auto nm2 = [&](const char *b, std::uint32_t i, std::uint32_t r) {
            return std::string(b) + "_" + std::to_string(i) + "_" + std::to_string(r);
        };

        {
            // State variable -> index of its jet among the LDS-resident input jets (0xffff: not an input of the clusters).
            std::vector<std::uint32_t> sv_ej(n_eq, 0xffffu);
            for (std::uint32_t e2 = 0; e2 < n_ej; ++e2) {
                sv_ej[ej_slots[e2]] = e2;
            }
            emit_utbl("hy_sv_ej", sv_ej);
        }
        // ---- the descriptors of the clusters: byte offsets (inside a row of the input jets) of the external inputs, two
        // per word, and the slab slots of the outputs; loaded at the head of the tape loads of a round (a table of
        // (n_dq + 2) * n_clusters words shared by all the workgroups: L2) - as values which live for the whole kernel they
        // were spilled, and a scratch reload inside the order loop makes the wavefront wait for ALL its outstanding tape
        // loads (loads, stores and scratch accesses share one in-order counter).
        std::ostringstream dc;
        dc << "#define HY_EJ(off) (*(const double *)((const char *)ejet + (off)))\n";
        dc << "#define HY_EJW(off) (*(double *)((char *)ejet + (off)))\n";
        // Tape accesses as raw buffer instructions: lane offset in a VGPR, row offset in an SGPR - one instruction per access
        // (the flat form costs a 64-bit VALU addition per access and two SALU instructions per row).
        dc << "typedef unsigned hy_u2 __attribute__((ext_vector_type(2)));\n";
        dc << "typedef unsigned hy_u4 __attribute__((ext_vector_type(4)));\n";
        dc << "const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void *)tape, 0, "
           << static_cast<std::uint64_t>(n_tape) * order * ncp * 8u << ", 0x00020000);\n";
        dc << "#define HY_TLD(lo, so) __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(trs, (lo), (so), 0))\n";
        dc << "#define HY_TST(v, lo, so) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(hy_u2, (v)), trs, (lo), (so), 0)\n";
        // Interleaved tape (round 6): the coefficients of the two tape members of a cluster side by side,
        // tape[(row * n_clusters + cluster) * 2 + member] - ONE 16-byte load / store per cluster and row instead of two 8-byte ones.
        dc << "typedef double hy_dd2 __attribute__((ext_vector_type(2)));\n";
        dc << "#define HY_TLD2(lo, so) __builtin_bit_cast(hy_dd2, __builtin_amdgcn_raw_buffer_load_b128(trs, (lo), (so), 0))\n";
        dc << "#define HY_TST2(va, vb, lo, so) { hy_dd2 t2_; t2_[0] = (va); t2_[1] = (vb); __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hy_u4, t2_), trs, (lo), (so), 0); }\n";
        // Compact form (2 words per cluster instead of n_ext / 2 + 2) when the tables are affine: the inputs of a cluster are
        // the coordinates of two bodies whose jets sit at a constant distance from one another ([component][body] layout of
        // the input jets) and its outputs sit at constant distances too - then one offset per body and one output slot
        // describe the cluster, the rest are constants which fold into the offset fields of the LDS instructions (and the
        // three components of a body share ONE address addition per row instead of three).
        std::vector<std::uint32_t> ex_word(n_ext), ex_shift(n_ext), ex_delta(n_ext, 0u);
        std::uint32_t out_delta[3] = {0u, 0u, 0u};
        const auto oslot_of = [&](std::uint32_t c, int x) {
            return static_cast<std::uint32_t>(pl.slot_of[pl.clusters[c][pl.out_pos[static_cast<std::uint32_t>(x)]]]);
        };
        bool compact = n_ext == 6u;
        if (compact) {
            for (std::uint32_t side = 0; side < 2u && compact; ++side) {
                const auto xb = pp.de[0][side];
                for (std::uint32_t comp = 0; comp < 3u && compact; ++comp) {
                    const auto x = pp.de[comp][side];
                    const auto d0 = static_cast<std::int64_t>(ej_index[static_cast<std::size_t>(x) * nc])
                                    - static_cast<std::int64_t>(ej_index[static_cast<std::size_t>(xb) * nc]);
                    for (std::uint32_t c = 0; c < nc && compact; ++c) {
                        compact = static_cast<std::int64_t>(ej_index[static_cast<std::size_t>(x) * nc + c])
                                      - static_cast<std::int64_t>(ej_index[static_cast<std::size_t>(xb) * nc + c])
                                  == d0;
                    }
                    compact = compact && d0 >= 0;
                    ex_word[x] = 0;
                    ex_shift[x] = 16u * side;
                    ex_delta[x] = static_cast<std::uint32_t>(d0) * 8u;
                }
            }
            for (std::uint32_t q = 1; q < 3u && compact; ++q) {
                const auto d0 = static_cast<std::int64_t>(oslot_of(0, io_of[q])) - static_cast<std::int64_t>(oslot_of(0, io_of[0]));
                for (std::uint32_t c = 0; c < nc && compact; ++c) {
                    compact = static_cast<std::int64_t>(oslot_of(c, io_of[q])) - static_cast<std::int64_t>(oslot_of(c, io_of[0])) == d0;
                }
                compact = compact && d0 >= 0;
                out_delta[q] = static_cast<std::uint32_t>(d0);
            }
        }
        if (!compact) {
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                ex_word[x] = x / 2u;
                ex_shift[x] = 16u * (x % 2u);
                ex_delta[x] = 0;
            }
        }
        const auto n_dq = compact ? 1u : (n_ext + 1u) / 2u;  // words holding input offsets
        const auto n_dw = compact ? 2u : n_dq + 2u;           // words per cluster
        {
            std::vector<std::uint32_t> dsc(static_cast<std::size_t>(n_dw) * nc, 0u);
            for (std::uint32_t c = 0; c < nc; ++c) {
                if (compact) {
                    dsc[c] = (ej_index[static_cast<std::size_t>(pp.de[0][0]) * nc + c] * 8u)
                             | ((ej_index[static_cast<std::size_t>(pp.de[0][1]) * nc + c] * 8u) << 16);
                    dsc[static_cast<std::size_t>(nc) + c] = oslot_of(c, io_of[0]);
                } else {
                    for (std::uint32_t x = 0; x < n_ext; ++x) {
                        dsc[static_cast<std::size_t>(x / 2u) * nc + c]
                            |= (ej_index[static_cast<std::size_t>(x) * nc + c] * 8u) << (16u * (x % 2u));
                    }
                    dsc[static_cast<std::size_t>(n_dq) * nc + c] = oslot_of(c, io_of[0]) | (oslot_of(c, io_of[1]) << 16);
                    dsc[static_cast<std::size_t>(n_dq + 1u) * nc + c] = oslot_of(c, io_of[2]);
                }
            }
            tbl << "__device__ const unsigned hy_dsc[" << dsc.size() << "] = {";
            for (const auto x : dsc) {
                tbl << x << ",";
            }
            tbl << "};\n";
        }
        // The byte offset of input x / the slab slot of output q of the lane's cluster in round r, as expressions.
        const auto ex_expr = [&](std::uint32_t x, std::uint32_t r) {
            const auto w = "dq_" + S(ex_word[x]) + "_" + S(r);
            auto e = ex_shift[x] == 0u ? "(" + w + " & 0xffffu)" : "(" + w + " >> 16)";
            if (ex_delta[x] != 0u) {
                e = "(" + e + " + " + U(ex_delta[x]) + ")";
            }
            return e;
        };
        const auto out_expr = [&](std::uint32_t q, std::uint32_t r) -> std::string {
            if (compact) {
                return "(dq_1_" + S(r) + " + " + U(out_delta[q]) + ")";
            }
            if (q == 0u) {
                return "(dq_" + S(n_dq) + "_" + S(r) + " & 0xffffu)";
            }
            return q == 1u ? "(dq_" + S(n_dq) + "_" + S(r) + " >> 16)" : "dq_" + S(n_dq + 1u) + "_" + S(r);
        };
        for (std::uint32_t r = 0; r < R; ++r) {
            const bool full = static_cast<std::uint64_t>(r + 1u) * bs <= nc;
            if (!full) {
                dc << "const bool live_" << r << " = tid + " << r * bs << "u < " << nc << "u;\n";
            }
        }
        // Sums of one dependency level with different numbers of terms (63 accelerations terms per body arrive as sums of
        // 8 + 8 + ... + 7: seven shapes per level for nbody(64), six of them with 48 nodes on 256 lanes) run as ONE group of
        // the widest shape, the missing operands read a slot which holds 0.0: the pairwise tree of the padded sum adds
        // exact zeros where the short one has nothing (same value), and a level is 7 + 2 + 2 lane rounds instead of 21.
        struct merged_sums {
            std::vector<std::size_t> groups;
            std::vector<std::uint32_t> nodes;
            std::uint32_t nargs = 0;
        };
        std::map<std::uint32_t, merged_sums> merged;
        std::vector<char> group_merged(pl.groups.size(), 0);
        // ("unpacked": A/B harness - one table per operand position, eight 2-byte reads per node.)
        const bool packed_idx = !wide && bopt("unpacked", 0) == 0;
        std::set<std::uint32_t> packed_levels;
        {
            for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                const auto &n0 = p.nodes[pl.groups[g].nodes[0] - n_eq];
                bool all_var = n0.kind == func_kind::sum && n0.args.size() <= 8u;
                for (const auto &o : n0.args) {
                    all_var = all_var && is_var(o);
                }
                if (all_var) {
                    auto &m = merged[pl.groups[g].level];
                    m.groups.push_back(g);
                    m.nargs = std::max(m.nargs, static_cast<std::uint32_t>(n0.args.size()));
                    m.nodes.insert(m.nodes.end(), pl.groups[g].nodes.begin(), pl.groups[g].nodes.end());
                }
            }
            for (auto it = merged.begin(); it != merged.end();) {
                if (it->second.groups.size() < 2u) {
                    it = merged.erase(it);
                } else {
                    for (const auto g : it->second.groups) {
                        group_merged[g] = 1;
                    }
                    const auto base = "hy_gm" + std::to_string(it->first);
                    // Order of the nodes of the level = lane assignment: lane l of a round gathers ITS operands from the slab
                    // with 64-bit reads, 32 lanes at a time, one cycle per distinct address sharing a bank. In the order of the
                    // decomposition (the three coordinates of a sum on adjacent lanes, 2016 slots = 0 mod 32 apart; partners in
                    // triangular order) a gather of the widest level of nbody(64) takes 4.9 cycles per 32 lanes; dealing the
                    // nodes greedily into groups of 32 with at most `tol` addresses per bank and operand position brings that
                    // to 2.3 (tol = 2). ("noglueperm": A/B harness.)
                    if (bopt("noglueperm", 0) == 0 && it->second.nodes.size() > 64u) {
                        auto &nodes = it->second.nodes;
                        const auto nargs = it->second.nargs;
                        const auto slots_of = [&](std::uint32_t u) {
                            std::vector<std::uint32_t> v(nargs, n_slots + 1u);
                            const auto &args = p.nodes[u - n_eq].args;
                            for (std::uint32_t a2 = 0; a2 < nargs && a2 < args.size(); ++a2) {
                                v[a2] = static_cast<std::uint32_t>(pl.slot_of[args[a2].idx]);
                            }
                            return v;
                        };
                        const auto score = [&](const std::vector<std::uint32_t> &ord) {
                            std::uint64_t tot = 0;
                            for (std::size_t g0 = 0; g0 < ord.size(); g0 += 32u) {
                                for (std::uint32_t a2 = 0; a2 < nargs; ++a2) {
                                    std::map<std::uint32_t, std::set<std::uint32_t>> bank;
                                    for (std::size_t j = g0; j < std::min(ord.size(), g0 + 32u); ++j) {
                                        const auto sl = slots_of(ord[j])[a2];
                                        bank[sl % 32u].insert(sl);
                                    }
                                    std::size_t mx = 0;
                                    for (const auto &[b_, st] : bank) {
                                        mx = std::max(mx, st.size());
                                    }
                                    tot += mx;
                                }
                            }
                            return tot;
                        };
                        auto best = nodes;
                        auto best_score = score(nodes);
                        for (std::uint32_t tol = 1; tol <= 3u; ++tol) {
                            std::vector<std::uint32_t> rem = nodes, ord;
                            while (!rem.empty()) {
                                std::vector<std::uint32_t> grp, keep;
                                std::vector<std::map<std::uint32_t, std::set<std::uint32_t>>> bank(nargs);
                                for (const auto u : rem) {
                                    bool ok = grp.size() < 32u;
                                    std::vector<std::uint32_t> sl;
                                    if (ok) {
                                        sl = slots_of(u);
                                        for (std::uint32_t a2 = 0; a2 < nargs && ok; ++a2) {
                                            const auto bi = bank[a2].find(sl[a2] % 32u);
                                            ok = bi == bank[a2].end() || bi->second.count(sl[a2]) != 0u || bi->second.size() < tol;
                                        }
                                    }
                                    if (ok) {
                                        grp.push_back(u);
                                        for (std::uint32_t a2 = 0; a2 < nargs; ++a2) {
                                            bank[a2][sl[a2] % 32u].insert(sl[a2]);
                                        }
                                    } else {
                                        keep.push_back(u);
                                    }
                                }
                                std::size_t taken = 0;
                                while (grp.size() < 32u && taken < keep.size()) {
                                    grp.push_back(keep[taken++]);
                                }
                                keep.erase(keep.begin(), keep.begin() + static_cast<std::ptrdiff_t>(taken));
                                ord.insert(ord.end(), grp.begin(), grp.end());
                                rem.swap(keep);
                            }
                            if (const auto sc = score(ord); sc < best_score) {
                                best_score = sc;
                                best = ord;
                            }
                        }
                        nodes = best;
                    }
                    if (packed_idx && it->second.nargs == 8u) {
                        // (The eight operand slots of a node side by side: ONE 16-byte table read per node instead of eight.)
                        std::vector<std::uint32_t> v;
                        for (const auto u : it->second.nodes) {
                            const auto &args = p.nodes[u - n_eq].args;
                            for (std::uint32_t a2 = 0; a2 < 8u; ++a2) {
                                v.push_back(a2 < args.size() ? static_cast<std::uint32_t>(pl.slot_of[args[a2].idx]) : n_slots + 1u);
                            }
                        }
                        emit_utbl(base + "_p", v);
                        packed_levels.insert(it->first);
                    } else {
                    for (std::uint32_t a2 = 0; a2 < it->second.nargs; ++a2) {
                        std::vector<std::uint32_t> v;
                        for (const auto u : it->second.nodes) {
                            const auto &args = p.nodes[u - n_eq].args;
                            v.push_back(a2 < args.size() ? static_cast<std::uint32_t>(pl.slot_of[args[a2].idx]) : n_slots + 1u);
                        }
                        emit_utbl(base + "_a" + std::to_string(a2), v);
                    }
                    }
                    std::vector<std::uint32_t> v;
                    for (const auto u : it->second.nodes) {
                        v.push_back(static_cast<std::uint32_t>(pl.slot_of[u]));
                    }
                    emit_utbl(base + "_o", v);
                    ++it;
                }
            }
        }
        // The same for the groups which stay on their own (<= 3 variable operands: operands and result in one 8-byte
        // record) and for the state-variable definitions (kind, slot, index of the input jet).
        std::set<std::size_t> q_groups;
        if (packed_idx) {
            for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                if (group_merged[g] != 0) {
                    continue;
                }
                const auto &n0 = p.nodes[pl.groups[g].nodes[0] - n_eq];
                std::vector<std::size_t> va;
                for (std::size_t a = 0; a < n0.args.size(); ++a) {
                    if (is_var(n0.args[a])) {
                        va.push_back(a);
                    }
                }
                if (va.empty() || va.size() > 3u) {
                    continue;
                }
                std::vector<std::uint32_t> v;
                for (const auto u : pl.groups[g].nodes) {
                    for (std::size_t q = 0; q < 3u; ++q) {
                        v.push_back(q < va.size() ? static_cast<std::uint32_t>(pl.slot_of[p.nodes[u - n_eq].args[va[q]].idx]) : 0u);
                    }
                    v.push_back(static_cast<std::uint32_t>(pl.slot_of[u]));
                }
                emit_utbl("hy_g" + std::to_string(g) + "_q", v);
                q_groups.insert(g);
            }
            std::vector<std::uint32_t> v;
            for (std::uint32_t i = 0; i < n_eq; ++i) {
                const auto &d = p.sv_defs[i];
                v.push_back(d.type == operand::kind::uvar ? 0u : (d.type == operand::kind::num ? 1u : 2u));
                v.push_back(d.type == operand::kind::uvar ? static_cast<std::uint32_t>(pl.slot_of[d.idx])
                                                          : (d.type == operand::kind::num ? 0u : d.idx));
                std::uint32_t ee = 0xffffu;
                for (std::uint32_t e2 = 0; e2 < n_ej; ++e2) {
                    if (ej_slots[e2] == i) {
                        ee = e2;
                    }
                }
                v.push_back(ee);
                v.push_back(0u);
            }
            emit_utbl("hy_sv_q", v);
        }
        // The LAST glue level fused into the state recursion (orders >= 1): when every node of the level is a difference of
        // two variables or a negation which only a state-variable definition reads (the accelerations of an N-body system
        // with equal masses: sum over the later partners minus sum over the earlier ones), the lane of the state variable
        // reads the two operands and subtracts itself - one barrier-separated phase per order less. Record per state
        // variable: operand slots a, b (b: the zero slot when unused), operation (0 a - 0.0, 2 nothing: constant
        // definition, 4 -a), index of the input jet. ("nofuse": A/B harness.)
        bool fuse_last = packed_idx && bopt("nofuse", 0) == 0 && pl.max_level >= 2u;
        std::vector<std::uint32_t> sv_f;
        if (fuse_last) {
            std::vector<char> read_elsewhere(p.nodes.size() + n_eq, 0);
            for (const auto &n : p.nodes) {
                for (const auto &o : n.args) {
                    if (is_var(o)) {
                        read_elsewhere[o.idx] = 1;
                    }
                }
            }
            for (const auto u : p.ev_u) {
                read_elsewhere[u] = 1;
            }
            std::map<std::uint32_t, std::uint32_t> n_defs;
            for (const auto &d : p.sv_defs) {
                if (d.type == operand::kind::uvar) {
                    ++n_defs[d.idx];
                }
            }
            for (std::size_t g = 0; g < pl.groups.size() && fuse_last; ++g) {
                if (pl.groups[g].level != pl.max_level) {
                    continue;
                }
                for (const auto u : pl.groups[g].nodes) {
                    const auto &n = p.nodes[u - n_eq];
                    const bool sub2 = n.kind == func_kind::sub && n.args.size() == 2u && is_var(n.args[0]) && is_var(n.args[1]);
                    const bool neg = n.kind == func_kind::prod && n.args.size() == 2u && n.args[0].type == operand::kind::num
                                     && n.args[0].value == -1. && is_var(n.args[1]);
                    fuse_last = fuse_last && (sub2 || neg) && read_elsewhere[u] == 0 && n_defs[u] == 1u;
                }
            }
        }
        if (fuse_last) {
            for (std::uint32_t i = 0; i < n_eq; ++i) {
                const auto &d = p.sv_defs[i];
                std::uint32_t a = n_slots + 1u, b = n_slots + 1u, op = 2u, ee = 0xffffu;
                if (d.type == operand::kind::uvar) {
                    op = 0u;
                    a = static_cast<std::uint32_t>(pl.slot_of[d.idx]);
                    if (d.idx >= n_eq && pl.lvl[d.idx] == pl.max_level && pl.cluster_of[d.idx] < 0) {
                        const auto &n = p.nodes[d.idx - n_eq];
                        if (n.kind == func_kind::sub) {
                            a = static_cast<std::uint32_t>(pl.slot_of[n.args[0].idx]);
                            b = static_cast<std::uint32_t>(pl.slot_of[n.args[1].idx]);
                        } else {
                            op = 4u;
                            a = static_cast<std::uint32_t>(pl.slot_of[n.args[1].idx]);
                        }
                    }
                }
                for (std::uint32_t e2 = 0; e2 < n_ej; ++e2) {
                    if (ej_slots[e2] == i) {
                        ee = e2;
                    }
                }
                sv_f.insert(sv_f.end(), {a, b, op, ee});
            }
            emit_utbl("hy_sv_f", sv_f);
        }
        v2_fuse_last = fuse_last;
        // One phase for the state recursion (with the fused level): a lane reads the operands of ITS definition - possibly
        // the slot of another state variable (x' = v) - and writes the next coefficient of its own variable; with the slots
        // of the state variables double-buffered by the parity of the order (bank 1 behind the slab) the write cannot
        // overtake another lane's read and the barrier between the two halves goes. Only when nothing but the state
        // recursion reads state-variable slots from the slab. MEASURED: nothing (1.647e6 with, 1.649e6 without): off; "dbuf" switches it on.
        bool dbuf = fuse_last && bopt("dbuf", 0) != 0;
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            dbuf = dbuf && ext_used[x] != 0;
        }
        for (std::size_t i = 0; i < p.nodes.size() && dbuf; ++i) {
            if (pl.cluster_of[n_eq + i] < 0) {
                for (const auto &o : p.nodes[i].args) {
                    dbuf = dbuf && !(is_var(o) && o.idx < n_eq);
                }
            }
        }
        v2_dbuf = dbuf;
        // LDS copies of the index tables read at every order (cluster descriptors, glue operands / results, state-variable
        // definitions), as far as they fit next to the slab and the input jets: a table lookup in front of every LDS access
        // of the glue phases and of the head of a round is a ~1 us round trip to L2 per dependency level when it is a
        // global load, ~0.1 us from LDS. Filled once per workgroup (the workgroups are persistent).
        {
            std::uint64_t lds_used = (static_cast<std::uint64_t>(n_slots) + 2u + (v2_dbuf ? n_eq : 0u)) * 8u + static_cast<std::uint64_t>(n_ej) * order * 8u
                                     + 3u * 8u * (bs / 64u) + 64u;
            const std::uint64_t lds_cap = 160u * 1024u - 256u;
            std::ostringstream cp, defs;
            const auto mirror = [&](const std::string &name, std::size_t n, const char *type, std::uint64_t esz) {
                if (lds_used + n * esz + 8u > lds_cap) {
                    return;
                }
                lds_used += (n * esz + 15u) / 16u * 16u;
                dc << "__shared__ " << type << " __attribute__((aligned(16))) l_" << name << "[" << n << "];\n";
                cp << "for (unsigned i_ = tid; i_ < " << n << "u; i_ += " << bs << "u) l_" << name << "[i_] = " << name
                   << "[i_];\n";
                defs << "#define " << name << " l_" << name << "\n";
            };
            {
                mirror("hy_dsc", static_cast<std::size_t>(n_dw) * nc, "unsigned", 4u);
                for (const auto &[name, n] : utbl_list) {
                    bool unused = false;
                    for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                        unused = unused || (group_merged[g] != 0 && name.rfind("hy_g" + std::to_string(g) + "_", 0) == 0);
                        unused = unused || (q_groups.count(g) != 0u && name.rfind("hy_g" + std::to_string(g) + "_", 0) == 0
                                            && name != "hy_g" + std::to_string(g) + "_q");
                    }
                    unused = unused || (packed_idx && (name == "hy_sv_kind" || name == "hy_sv_idx" || name == "hy_sv_ej"));
                    if (!unused && (name.rfind("hy_g", 0) == 0 || name.rfind("hy_sv_", 0) == 0)) {
                        mirror(name, n, "unsigned short", 2u);
                    }
                }
            }
            dc << cp.str() << "__syncthreads();\n" << defs.str();
        }
        v2_decl = dc.str();
        // Cluster index of the lane in round r and its byte offset inside a tape row; the descriptor loads.
        const auto desc_loads = [&](std::uint32_t r) {
            const bool full = static_cast<std::uint64_t>(r + 1u) * bs <= nc;
            // (The empty asm statement keeps the loads where they are: as loop invariants they would be hoisted out of the
            // order loop and spilled.)
            os << "unsigned cl_" << r << " = "
               << (full ? "tid + " + U(r * bs) : "live_" + S(r) + " ? tid + " + U(r * bs) + " : " + U(nc - 1u)) << ";\n";
            os << "asm volatile(\"\" : \"+v\"(cl_" << r << "));\n";
            os << "const unsigned lo_" << r << " = cl_" << r << " * 8u;\n";
            for (std::uint32_t j = 0; j < n_dw; ++j) {
                os << "const unsigned dq_" << j << "_" << r << " = hy_dsc[" << U(static_cast<std::uint64_t>(j) * nc) << " + cl_" << r
                   << "];\n";
            }
        };

        const auto cl_expr = [&](std::uint32_t r) {
            const bool full = static_cast<std::uint64_t>(r + 1u) * bs <= nc;
            return full ? "tid + " + U(r * bs) : "(live_" + S(r) + " ? tid + " + U(r * bs) + " : " + U(nc - 1u) + ")";
        };
        const auto unpack = [&](std::uint32_t r) {
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                os << "const unsigned ex" << x << " = " << ex_expr(x, r) << ";\n";
            }
            os << "const unsigned dvo0 = " << out_expr(0, r) << ", dvo1 = " << out_expr(1, r) << ", dvo2 = " << out_expr(2, r)
               << ";\n";
        };
        // d_c of the row at byte offset `row` (an expression): minuend - subtrahend (src/detail/sub.cpp:60-124).
        const auto diff = [&](const std::string &name, std::uint32_t c, const std::string &row) {
            os << "const double " << name << " = HY_EJ(" << row << " + ex" << pp.de[c][0] << ") - HY_EJ(" << row << " + ex"
               << pp.de[c][1] << ");\n";
        };
        const auto store_out = [&](std::uint32_t, const std::string v[3]) {
            os << "slab[dvo0] = " << v[0] << ";\n";
            os << "slab[dvo1] = " << v[1] << ";\n";
            os << "slab[dvo2] = " << v[2] << ";\n";
        };
        // NOTE: the idle lanes of a partial last round work on a copy of the last cluster and store the same values to the
        // same places as the lane which owns it: no predicated stores, i.e. no divergent control flow in the cluster phase
        // (see the toolchain note on exec-masked regions under register pressure, heyoka_amd/codegen_check.py).
        const auto is_full = [&](std::uint32_t) { return true; };

        os << "double m0 = 0.0, mo = 0.0, mom1 = 0.0;\n";
        for (std::uint32_t r = 0; r < sv_rounds; ++r) {
            os << "{ const unsigned i = tid + " << r * bs << "u; if (i < " << n_eq
               << "u) m0 = hy_max(m0, fabs(slab[i])); }\n";
        }
        for (std::uint32_t r = 0; r < R; ++r) {
            for (std::uint32_t m = 0; m < M; ++m) {
                os << "double " << nm2("ca", m, r) << " = 0.0, " << nm2("cb", m, r) << " = 0.0;\n";
            }
            if (recip) {
                os << "double rca_" << r << " = 0.0;\n";
            }
        }

        // ---- order 0: the node rules of the other modes ----
        os << "// ---- order 0 ----\n";
        for (std::uint32_t r = 0; r < R; ++r) {
            os << "{\n";
            desc_loads(r);
            unpack(r);
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                e.val(pl.ext_u[0][x], 0) = e.def("HY_EJ(" + U(ej0b) + " + ex" + S(x) + ")");
            }
            for (const auto u : t0) {
                e.node(u - n_eq, 0);
            }
            os << nm2("ca", 0, r) << " = " << e.val(t0[pp.sq], 0) << ";\n";
            os << nm2("cb", 0, r) << " = " << e.val(t0[pp.pw], 0) << ";\n";
            if (recip) {
                os << "rca_" << r << " = 1.0 / " << nm2("ca", 0, r) << ";\n";
            }
            const std::string v[3] = {e.val(t0[pp.pr[0]], 0), e.val(t0[pp.pr[1]], 0), e.val(t0[pp.pr[2]], 0)};
            if (!is_full(r)) {
                os << "if (live_" << r << ") {\n";
            }
            store_out(r, v);
            if (!is_full(r)) {
                os << "}\n";
            }
            os << "}\n";
        }
        // The glue groups of a dependency level, STAGED: the nodes of a level do not read each other, but the compiler
        // cannot know that the slab store of one block does not alias the slab loads of the next one - emitted block by block
        // (index lookup, operands, sum, store) a level is one chain of ~40 dependent LDS round trips on the single wavefront
        // of a SIMD. Here: all the index lookups of the level, then all the operand loads, then the arithmetic, then the
        // stores.
        const auto glue_level_staged = [&](std::uint32_t lev, std::uint32_t k) {
            std::string st_idx, st_ld, st_ar, st_wr;
            const auto take = [&]() {
                auto t = os.str();
                os.str("");
                os.clear();
                return t;
            };
            const auto before = take();
            if (const auto mit = merged.find(lev); mit != merged.end() && !(exp_mode == 2 && k != 0u)) {
                const auto &m = mit->second;
                const auto gn = "hy_gm" + std::to_string(lev);
                const auto ng = static_cast<std::uint32_t>(m.nodes.size());
                for (std::uint32_t r = 0; r * bs < ng; ++r) {
                    const auto tag = "_m" + std::to_string(lev) + "_" + std::to_string(r) + "_" + std::to_string(k == 0u ? 0 : 1);
                    const bool partial = (r + 1u) * bs > ng;
                    const auto inside = "(tid + " + std::to_string(r * bs) + "u < " + std::to_string(ng) + "u)";
                    os << "const unsigned j" << tag << " = "
                       << (partial ? inside + " ? tid + " + std::to_string(r * bs) + "u : " + std::to_string(ng - 1u) + "u"
                                   : "tid + " + std::to_string(r * bs) + "u")
                       << ";\n";
                    if (packed_levels.count(lev) != 0u) {
                        os << "const hy_u4 iw" << tag << " = *(const hy_u4 *)(" << gn << "_p + 8u * j" << tag << ");\n";
                        for (std::uint32_t a2 = 0; a2 < 8u; ++a2) {
                            os << "const unsigned ia" << a2 << tag << " = " << ((a2 & 1u) != 0u ? "(iw" + tag + "[" + std::to_string(a2 / 2u) + "] >> 16)"
                                                                                           : "(iw" + tag + "[" + std::to_string(a2 / 2u) + "] & 0xffffu)")
                               << ";\n";
                        }
                    } else {
                    for (std::uint32_t a2 = 0; a2 < m.nargs; ++a2) {
                        os << "const unsigned ia" << a2 << tag << " = " << gn << "_a" << a2 << "[j" << tag << "];\n";
                    }
                    }
                    os << "const unsigned io" << tag << " = "
                       << (partial ? inside + " ? (unsigned)" + gn + "_o[j" + tag + "] : " + std::to_string(n_slots) + "u"
                                   : "(unsigned)" + gn + "_o[j" + tag + "]")
                       << ";\n";
                    st_idx += take();
                    std::vector<std::string> terms;
                    for (std::uint32_t a2 = 0; a2 < m.nargs; ++a2) {
                        terms.push_back(e.def("slab[ia" + std::to_string(a2) + tag + "]"));
                    }
                    st_ld += take();
                    const auto res = e.pairwise_sum(std::move(terms));
                    st_ar += take();
                    st_wr += "slab[io" + tag + "] = " + res + ";\n";
                }
            }
            for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                const auto &grp = pl.groups[g];
                if (grp.level != lev || group_merged[g] != 0 || (exp_mode == 2 && k != 0u)) {
                    continue;
                }
                const auto rep = grp.nodes[0];
                const auto &n0 = p.nodes[rep - n_eq];
                const auto gn = "hy_g" + std::to_string(g);
                const auto ng = static_cast<std::uint32_t>(grp.nodes.size());
                for (std::uint32_t r = 0; r * bs < ng; ++r) {
                    const auto tag = "_" + std::to_string(g) + "_" + std::to_string(r) + "_" + std::to_string(k == 0u ? 0 : 1);
                    const bool partial = (r + 1u) * bs > ng;
                    os << "const unsigned j" << tag << " = ";
                    if (partial) {
                        os << "(tid + " << r * bs << "u < " << ng << "u) ? tid + " << r * bs << "u : " << ng - 1u << "u;\n";
                    } else {
                        os << "tid + " << r * bs << "u;\n";
                    }
                    if (q_groups.count(g) != 0u) {
                        os << "const hy_u2 iq" << tag << " = *(const hy_u2 *)(" << gn << "_q + 4u * j" << tag << ");\n";
                        std::uint32_t q = 0;
                        for (std::size_t a = 0; a < n0.args.size(); ++a) {
                            if (is_var(n0.args[a])) {
                                os << "const unsigned ia" << a << tag << " = "
                                   << (q == 0u ? "(iq" + tag + "[0] & 0xffffu)" : (q == 1u ? "(iq" + tag + "[0] >> 16)" : "(iq" + tag + "[1] & 0xffffu)"))
                                   << ";\n";
                                ++q;
                            }
                        }
                        os << "const unsigned io" << tag << " = "
                           << (partial ? "(tid + " + std::to_string(r * bs) + "u < " + std::to_string(ng) + "u) ? (iq" + tag + "[1] >> 16) : "
                                             + std::to_string(n_slots) + "u"
                                       : "(iq" + tag + "[1] >> 16)")
                           << ";\n";
                    } else {
                    for (std::size_t a = 0; a < n0.args.size(); ++a) {
                        if (is_var(n0.args[a])) {
                            os << "const unsigned ia" << a << tag << " = " << gn << "_a" << a << "[j" << tag << "];\n";
                        }
                    }
                    os << "const unsigned io" << tag << " = "
                       << (partial ? "(tid + " + std::to_string(r * bs) + "u < " + std::to_string(ng) + "u) ? (unsigned)" + gn
                                         + "_o[j" + tag + "] : " + std::to_string(n_slots) + "u"
                                   : "(unsigned)" + gn + "_o[j" + tag + "]")
                       << ";\n";
                    }
                    st_idx += take();
                    const auto saved = e.numpar_override;
                    std::vector<std::pair<std::uint32_t, std::string>> saved_vals;
                    for (std::size_t a = 0; a < n0.args.size(); ++a) {
                        const auto &o = n0.args[a];
                        if (is_var(o)) {
                            const auto nm_ = e.def(exp_mode == 5 ? "slab[(ia" + std::to_string(a) + tag + " & 1u) + tid + "
                                                                       + std::to_string(a * 2u) + "u]"
                                                                 : "slab[ia" + std::to_string(a) + tag + "]");
                            saved_vals.emplace_back(o.idx, e.val(o.idx, k));
                            e.val(o.idx, k) = nm_;
                        } else if (o.type == operand::kind::num) {
                            e.numpar_override[&o] = gn + "_a" + std::to_string(a) + "[j" + tag + "]";
                        }
                    }
                    st_ld += take();
                    if (n0.kind == func_kind::prod && n0.args[0].type == operand::kind::num && n0.args[0].value == -1.) {
                        e.numpar_override.erase(&n0.args[0]);
                    }
                    e.node(rep - n_eq, k);
                    st_ar += take();
                    st_wr += "slab[io" + tag + "] = " + e.val(rep, k) + ";\n";
                    for (auto it = saved_vals.rbegin(); it != saved_vals.rend(); ++it) {
                        e.val(it->first, k) = it->second;
                    }
                    e.numpar_override = saved;
                }
            }
            os << before << "{\n" << st_idx << st_ld << st_ar << st_wr << "}\n";
        };
        const auto glue_levels = [&](std::uint32_t k) {
            for (std::uint32_t lev = 1; lev <= pl.max_level; ++lev) {
                if (k != 0u && v2_fuse_last && lev == pl.max_level) {
                    continue; // (computed by the lanes of the state variables, below)
                }
                glue_level_staged(lev, k);
                sync();
            }
        };
        glue_levels(0);
        // State-variable recursion (src/taylor_02.cpp:245-287) with the order-(k + 1) values of the external inputs
        // recorded in LDS. dyn: k is the loop variable.
        const auto recursion = [&](bool dyn) {
            os << "{\n";
            for (std::uint32_t r = 0; r < sv_rounds; ++r) {
                os << "double xn" << r << " = 0.0;\nunsigned svej" << r << " = 0xffffu;\n";
                os << "{ const unsigned i = tid + " << r * bs << "u; if (i < " << n_eq << "u) {\n";
                if (dyn && v2_fuse_last) {
                    os << "const hy_u2 svf = *(const hy_u2 *)(hy_sv_f + 4u * i);\n";
                    os << "svej" << r << " = svf[1] >> 16;\n";
                    os << "const unsigned op_ = svf[1] & 0xffffu;\n";
                    if (v2_dbuf) {
                        // (Order-k coefficients of the state variables: bank (k - 1) & 1; the new ones go to bank k & 1.)
                        os << "const unsigned rbo = ((k - 1u) & 1u) * " << U(n_slots + 2u) << ";\n";
                        os << "const unsigned sa_ = svf[0] & 0xffffu, sb_ = svf[0] >> 16;\n";
                        os << "const double xa_ = slab[sa_ + (sa_ < " << n_eq << "u ? rbo : 0u)], xb_ = slab[sb_ + (sb_ < " << n_eq
                           << "u ? rbo : 0u)];\n";
                    } else {
                        os << "const double xa_ = slab[svf[0] & 0xffffu], xb_ = slab[svf[0] >> 16];\n";
                    }
                    os << "const double xv = op_ == 4u ? -xa_ : xa_ - xb_;\n";
                    if (v2_recip) {
                        os << "const double xq = xv * rkd1;\n";
                        os << "xn" << r << " = op_ == 2u ? 0.0 : __builtin_fma(__builtin_fma(-xq, kd + 1.0, xv), rkd1, xq);\n";
                    } else {
                        os << "xn" << r << " = op_ == 2u ? 0.0 : xv / (kd + 1.0);\n";
                    }
                    os << "} }\n";
                    continue;
                }
                if (packed_idx) {
                    os << "const hy_u2 svq = *(const hy_u2 *)(hy_sv_q + 4u * i);\n";
                    os << "svej" << r << " = svq[1] & 0xffffu;\n";
                    os << "const unsigned kd_ = svq[0] & 0xffffu, svi_ = svq[0] >> 16;\n";
                } else {
                    os << "const unsigned kd_ = hy_sv_kind[i], svi_ = hy_sv_idx[i];\n";
                }
                if (dyn) {
                    if (v2_recip) {
                        os << "if (kd_ == 0u) { const double xv = slab[svi_], xq = xv * rkd1; xn" << r
                           << " = __builtin_fma(__builtin_fma(-xq, kd + 1.0, xv), rkd1, xq); }\n";
                    } else {
                        os << "if (kd_ == 0u) xn" << r << " = slab[svi_] / (kd + 1.0);\n";
                    }
                } else {
                    os << "if (kd_ == 0u) xn" << r << " = slab[svi_];\n";
                    os << "else if (kd_ == 1u) xn" << r << " = hy_sv_val[i];\n";
                    os << "else xn" << r << " = a.pars[(u64)svi_ * N + s];\n";
                }
                os << "} }\n";
            }
            if (!(dyn && v2_dbuf)) {
                sync();
            }
            for (std::uint32_t r = 0; r < sv_rounds; ++r) {
                os << "{ const unsigned i = tid + " << r * bs << "u; if (i < " << n_eq << "u) {\n";
                if (dyn && v2_dbuf) {
                    os << "slab[i + (k & 1u) * " << U(n_slots + 2u) << "] = xn" << r << ";\n";
                } else {
                    os << "slab[i] = xn" << r << ";\n";
                }
                if (dyn) {
                    os << "sjet[(k + 1u) * " << n_eq << "u + i] = xn" << r << ";\n";
                    os << "const unsigned ee = " << (packed_idx ? "svej" + std::to_string(r) : std::string("hy_sv_ej[i]")) << ";\n";
                    os << "if (ee != 0xffffu && k + 1u < " << P << "u) HY_EJW((" << P - 2u << "u - k) * " << U(ejrow)
                       << " + ee * 8u) = xn" << r << ";\n";
                    os << "if (k + 1u == " << P << "u) mo = hy_max(mo, fabs(xn" << r << "));\n";
                    os << "else if (k + 2u == " << P << "u) mom1 = hy_max(mom1, fabs(xn" << r << "));\n";
                } else {
                    os << "sjet[" << n_eq << "u + i] = xn" << r << ";\n";
                    os << "const unsigned ee = " << (packed_idx ? "svej" + std::to_string(r) : std::string("hy_sv_ej[i]")) << ";\n";
                    os << "if (ee != 0xffffu) HY_EJW(" << U(static_cast<std::uint64_t>(P - 2u) * ejrow) << " + ee * 8u) = xn" << r
                       << ";\n";
                }
                os << "} }\n";
            }
            sync();
            os << "}\n";
        };
        recursion(false);

        // ---- orders 1 .. P - 1 ----
        // Tape registers of slot 1 (set 1 of the pipeline): slot 1 pairs order 1 with order k - 1, whose coefficients the lane
        // computed itself one order earlier - they are handed over in registers (no load: the head of a group does not
        // wait for the memory system). NOTE: loading them back right behind their stores is NOT an option: under load
        // such a load was observed to overtake the store (stale values, runaway step sizes).
        // Rounds per pipeline group (the accumulators, descriptors and tape registers of a group are live at the same
        // time: R rounds at once do not fit the register file next to the register copies of the low orders).
        std::uint32_t G = R;
        if (R > 4u) {
            G = 4;
        }
        if (v2_two_waves) {
            G = 2;
        }
        G = static_cast<std::uint32_t>(bopt("G", static_cast<int>(G)));
        G = std::max(1u, std::min(G, R));
        // Depth of the tape-load pipeline, in slots: the loads of slot i + D are issued at the head of slot i. One slot of a
        // group is G * ~80 instructions, i.e. a fraction of a microsecond, against 1 - 2 us for a load which misses L2.
        // (Measured on nbody(64), 65 536 systems, rows < 5 in registers: depth 1 / 2 / 3 / 4 with groups of two rounds =
        // 9.8e5 / 9.7e5 / 8.8e5 / 8.6e5 system-steps/s - the registers of a deeper pipeline cost more than the latency they
        // hide; the kernel is bound by the instruction issue of its single wavefront per SIMD.)
        const std::uint32_t D = static_cast<std::uint32_t>(std::max(1, bopt("D", v2_two_waves ? 2 : 1)));
        // Cross-group prefetch ("xpf"): the high rows of the slots 2 .. D + 1 of a group - rows of order <= k - 2, stored long
        // ago - are requested while the PREVIOUS group finishes (for the first group of an order: behind the last group of
        // the order before, across the glue phases), so that a group does not start with the latency of the memory system
        // exposed (D slots of two rounds are ~0.5 us of work against 1 - 2 us): registers xa<i>_<q> / xb<i>_<q>, q the
        // position of the round in its group, live from the end of a group to the slots of the next one.
        const bool xpf = bopt("xpf", 0) != 0 && D + 1u < M && D + 1u < T && !reg_high && exp_mode == 0 && R % G == 0u;
        const auto xname = [&](const char *b, std::uint32_t i, std::uint32_t q) { return std::string(b) + S(i) + "_" + S(q); };
        const std::uint32_t hand = (!xpf && !reg_high && exp_mode == 0)
                                       ? std::min<std::uint32_t>(static_cast<std::uint32_t>(std::max(1, bopt("hand", static_cast<int>(hand_dflt)))), std::min(T - 1u, il_tape ? T - 1u : M - 1u))
                                       : 1u;
        const bool hand2 = hand >= 2u;
        v2_hand = hand;
        if (xpf) {
            for (std::uint32_t i = 2; i <= D + 1u; ++i) {
                for (std::uint32_t q = 0; q < G; ++q) {
                    os << "double " << xname("xa", i, q) << " = 0.0, " << xname("xb", i, q) << " = 0.0;\n";
                }
            }
        }
        for (std::uint32_t r = 0; r < R; ++r) {
            os << "double ap1_" << r << " = 0.0, bp1_" << r << " = 0.0;\n";
            // ("hand=H": the coefficients of the orders k - 2 ... k - H handed over in registers too: the slots 2 ... H need no
            // tape load for their high member.)
            for (std::uint32_t h = 2; h <= hand; ++h) {
                os << "double ah" << h << "_" << r << " = 0.0, bh" << h << "_" << r << " = 0.0;\n";
            }
        }
        // ("hoist", A/B harness: the descriptors of the clusters of a lane loaded once per step, in registers through the orders.)
        const bool hoist = bopt("hoist", v2_two_waves && !v2_lean ? 1 : 0) != 0;
        // ("keepdz": the order-0 differences of the clusters of a lane in registers through the orders - they are read
        // twice per order and round otherwise.)
        const bool keepdz = bopt("keepdz", v2_two_waves && !v2_lean ? 1 : 0) != 0;
        if (hoist) {
            for (std::uint32_t r = 0; r < R; ++r) {
                desc_loads(r);
            }
        }
        if (keepdz) {
            for (std::uint32_t r = 0; r < R; ++r) {
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    os << "double hdz" << c << "_" << r << ";\n";
                }
                os << "{\n";
                if (!hoist) {
                    desc_loads(r);
                }
                for (std::uint32_t x = 0; x < n_ext; ++x) {
                    os << "const unsigned ex" << x << " = " << ex_expr(x, r) << ";\n";
                }
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    os << "hdz" << c << "_" << r << " = HY_EJ(" << U(ej0b) << " + ex" << pp.de[c][0] << ") - HY_EJ(" << U(ej0b) << " + ex"
                       << pp.de[c][1] << ");\n";
                }
                os << "}\n";
            }
        }
        os << "#pragma nounroll\nfor (unsigned k = 1; k < " << P << "u; ++k) {\n";
        os << "const double kd = (double)k;\n";
        os << "const double c0k = kd * " << fp_literal(ex) << ";\n";
        if (recip) {
            os << "const double rkd = 1.0 / kd, rkd1 = 1.0 / (kd + 1.0);\n";
        }
        os << "const unsigned nm1 = k >> 1;\nconst bool even = (k & 1u) == 0u;\n";
        os << "const unsigned kbr = (" << P - 1u << "u - k) * " << U(ejrow) << ";\n";
        if (il_tape) {
            os << "const unsigned tpa = k * " << 2u * rowb << "u;\n";
        } else {
            os << "const unsigned tpa = (" << arow0 << "u + k) * " << rowb << "u;\n";
            os << "const unsigned tpb = (" << brow0 << "u + k) * " << rowb << "u;\n";
        }
        // The cluster phase of order k, SLOT-major: for slot i = 1 .. floor(k / 2) (early exit after the last one) the
        // rounds 0 .. R - 1 of the lane, with the five accumulators of every round in registers. One software pipeline runs
        // through the whole (slot, round) sequence of an order:
        //  * tape loads: those of slot i + 1 (all rounds) are issued, unconditionally, at the head of slot i - rows of
        //    slots beyond the last one are clamped to rows which are loaded anyway (L1 hits) -, so that an order pays the
        //    latency of the memory system once (slot 1) and not once per round;
        //  * LDS reads: the 12 raw values of a step are read before the arithmetic of the previous step (two sets,
        //    ping-pong): with ONE wavefront per SIMD nothing else covers the LDS latency.
        const auto raw = [&](std::uint32_t set, std::uint32_t q, std::uint32_t c, std::uint32_t side) {
            return "w" + S(set) + "_" + S(q) + S(c) + S(side); // q: 0 = row i, 1 = row p
        };
        const auto unpack_ex = [&](std::uint32_t r) {
            for (std::uint32_t x = 0; x < n_ext; ++x) {
                os << "const unsigned ex" << x << " = " << ex_expr(x, r) << ";\n";
            }
        };
        // The LDS reads of step (i, r) into set `set` (inside their own scope: the unpacked offsets are temporaries).
        const auto lds_step = [&](std::uint32_t i, std::uint32_t r, std::uint32_t set) {
            if (exp_mode == 3 || exp_mode == 34) {
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    for (std::uint32_t side = 0; side < 2u; ++side) {
                        os << raw(set, 0, c, side) << " = kd;\n" << raw(set, 1, c, side) << " = c0k;\n";
                    }
                }
                return;
            }
            // (eo*_r: byte offsets of the inputs of the lane's cluster of round r inside a row; kx*_r: the same plus the row of
            // order k - one register each for the duration of a group, so that every read is "register + constant".)
            for (std::uint32_t c = 0; c < 3u; ++c) {
                for (std::uint32_t side = 0; side < 2u; ++side) {
                    os << raw(set, 0, c, side) << " = HY_EJ(eo" << pp.de[c][side] << "_" << r << " + "
                       << U(static_cast<std::uint64_t>(P - 1u - i) * ejrow) << ");\n";
                    os << raw(set, 1, c, side) << " = HY_EJ(kx" << pp.de[c][side] << "_" << r << " + "
                       << U(static_cast<std::uint64_t>(i) * ejrow) << ");\n";
                }
            }
        };
        // The tape loads of slot i (rounds r0 .. r1 - 1) into the set i % 2.
        const auto gname = [&](const char *b, std::uint32_t i, std::uint32_t r) {
            return std::string(b) + S(i % (D + 1u)) + "_" + S(r);
        };
        const auto tape_loads = [&](std::uint32_t i, std::uint32_t r0, std::uint32_t r1) {
            const bool skip_high = i >= 2u && i <= hand;
            if (skip_high && i < M) {
                return;
            }
            const auto back = "(nm1 < " + S(i) + "u ? nm1 : " + S(i) + "u) * " + std::to_string(rowb) + "u";
            const auto own = "(k < " + S(i) + "u ? k : " + S(i) + "u) * " + std::to_string(rowb) + "u";
            if (exp_mode == 4 || exp_mode == 34) {
                for (std::uint32_t r = r0; r < r1; ++r) {
                    os << gname("ap", i, r) << " = kd;\n" << gname("bp", i, r) << " = c0k;\n";
                    if (i >= M) {
                        os << gname("al", i, r) << " = kd;\n" << gname("bl", i, r) << " = c0k;\n";
                    }
                }
                return;
            }
            // (reg_high: no request while the high member of the slot is a row in registers.)
            os << (reg_high && i < M ? "if (k >= " + S(i + M) + "u) {\n" : std::string("{\n"));
            if (il_tape) {
                os << "const unsigned sa = tpa - 2u * " << back << ";\n";
                if (i >= M) {
                    os << "const unsigned oa = 2u * " << own << ";\n";
                }
                for (std::uint32_t r = r0; r < r1; ++r) {
                    if (!skip_high) {
                        os << "{ const hy_dd2 t_ = HY_TLD2(2u * lo_" << r << ", sa); " << gname("ap", i, r) << " = t_[0]; "
                           << gname("bp", i, r) << " = t_[1]; }\n";
                    }
                    if (i >= M) {
                        os << "{ const hy_dd2 t_ = HY_TLD2(2u * lo_" << r << ", oa); " << gname("al", i, r) << " = t_[0]; "
                           << gname("bl", i, r) << " = t_[1]; }\n";
                    }
                }
                os << "}\n";
                return;
            }
            os << "const unsigned sa = tpa - " << back << ", sb = tpb - " << back << ";\n";
            if (i >= M) {
                os << "const unsigned oa = " << arow0 * rowb << "u + " << own << ", ob = " << brow0 * rowb << "u + " << own << ";\n";
            }
            for (std::uint32_t r = r0; r < r1; ++r) {
                os << gname("ap", i, r) << " = HY_TLD(lo_" << r << ", sa);\n";
                os << gname("bp", i, r) << " = HY_TLD(lo_" << r << ", sb);\n";
                if (i >= M) {
                    os << gname("al", i, r) << " = HY_TLD(lo_" << r << ", oa);\n";
                    os << gname("bl", i, r) << " = HY_TLD(lo_" << r << ", ob);\n";
                }
            }
            os << "}\n";
        };
        const auto emit_group = [&](std::uint32_t r0, std::uint32_t r1, std::uint32_t gi) {
            const auto Rg = r1 - r0;
            // ("nopp", A/B harness: no ping-pong - the LDS reads of a step right in front of its arithmetic, one register set.)
            const bool nopp = bopt("nopp", v2_two_waves ? 1 : 0) != 0;
            const auto step_set = [&](std::uint32_t i, std::uint32_t r) { return nopp ? 0u : ((i - 1u) * Rg + (r - r0)) % 2u; };
            os << "{\n";
            // Declarations of the pipeline registers.
            for (std::uint32_t set = 0; set <= D; ++set) {
                for (std::uint32_t r = r0; r < r1; ++r) {
                    std::string d;
                    if (set != 1u) {
                        d = "ap" + S(set) + "_" + S(r) + " = 0.0, bp" + S(set) + "_" + S(r) + " = 0.0";
                    }
                    if (M < T) {
                        d += std::string(d.empty() ? "" : ", ") + "al" + S(set) + "_" + S(r) + " = 0.0, bl" + S(set) + "_" + S(r)
                             + " = 0.0";
                    }
                    if (!d.empty()) {
                        os << "double " << d << ";\n";
                    }
                }
            }
            for (std::uint32_t set = 0; set < 2u; ++set) {
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    for (std::uint32_t side = 0; side < 2u; ++side) {
                        os << "double " << raw(set, 0, c, side) << " = 0.0, " << raw(set, 1, c, side) << " = 0.0;\n";
                    }
                }
            }
            // Head: descriptors, the tape loads of slot 1, slot 0 (order-k differences x order-0 ones).
            for (std::uint32_t r = r0; r < r1; ++r) {
                if (!hoist) {
                    desc_loads(r);
                }
                for (std::uint32_t x = 0; x < n_ext; ++x) {
                    // (The optimiser re-derives these from the packed word where they are used - one SDWA addition per read
                    // of a dynamic row; pinning them in registers was measured: they end up in AGPRs and every read pays a
                    // copy instead.)
                    os << "const unsigned eo" << x << "_" << r << " = " << ex_expr(x, r) << ";\n";
                    os << "const unsigned kx" << x << "_" << r << " = kbr + eo" << x << "_" << r << ";\n";
                }
            }
            if (!xpf) {
                for (std::uint32_t i = 2; i <= D && i < T; ++i) {
                    tape_loads(i, r0, r1);
                }
            }
            if (hand2) {
                for (std::uint32_t r = r0; r < r1; ++r) {
                    os << "const double ah1s_" << r << " = ap1_" << r << ", bh1s_" << r << " = bp1_" << r << ";\n";
                }
            }
            for (std::uint32_t r = r0; r < r1; ++r) {
                os << "double ssq_" << r << ", spw_" << r << " = 0.0, sf0_" << r << ", sf1_" << r << ", sf2_" << r << ";\n";
                os << "{\n";
                unpack_ex(r);
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    diff("dk" + S(c), c, "kbr");
                    if (keepdz) {
                        os << "const double dz" << c << " = hdz" << c << "_" << r << ";\n";
                    } else {
                        diff("dz" + S(c), c, U(ej0b));
                    }
                }
                os << "ssq_" << r << " = dz0 * dk0;\nssq_" << r << " = __builtin_fma(dz1, dk1, ssq_" << r << ");\nssq_" << r
                   << " = __builtin_fma(dz2, dk2, ssq_" << r << ");\n";
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    os << "sf" << c << "_" << r << " = dk" << c << " * " << nm2("cb", 0, r) << ";\n";
                }
                os << "}\n";
            }
            if (T > 1u) {
                if (!nopp) {
                    lds_step(1, r0, step_set(1, r0));
                    os << "__builtin_amdgcn_sched_barrier(0);\n";
                }
                os << (exp_mode == 1 ? "if (nm1 > 1000u) {\n" : "if (nm1 != 0u) {\n");
                for (std::uint32_t i = 1; i < T; ++i) {
                    os << "{\n";
                    if (i + D < T && !(xpf && i + D <= D + 1u)) {
                        tape_loads(i + D, r0, r1);
                    }
                    // Weights of the last slot of an even order (middle term): 1/2 on the squares, 0 on the mirrored
                    // products; scalar factors k a - j (a + 1) of the terms j = i and j = p = k - i.
                    os << "const bool mid = even & (nm1 == " << i << "u);\n";
                    os << "const double wq = mid ? 0.5 : 1.0, w2 = mid ? 0.0 : 1.0;\n";
                    os << "const double ci = c0k - " << fp_literal(static_cast<double>(i) * e1) << ";\n";
                    os << "const double cp = c0k - (kd - " << fp_literal(static_cast<double>(i)) << ") * " << fp_literal(e1)
                       << ";\n";
                    if (reg_high_br && i >= 2u && i < M) {
                        os << "if (k < " << S(i + M) << "u) {\n";
                        for (std::uint32_t r = r0; r < r1; ++r) {
                            std::string a = nm2("ca", M - 1u, r), b = nm2("cb", M - 1u, r);
                            for (std::uint32_t m = M - 1u; m-- > i;) {
                                a = "(k == " + S(i + m) + "u ? " + nm2("ca", m, r) + " : " + a + ")";
                                b = "(k == " + S(i + m) + "u ? " + nm2("cb", m, r) + " : " + b + ")";
                            }
                            os << gname("ap", i, r) << " = " << a << ";\n" << gname("bp", i, r) << " = " << b << ";\n";
                        }
                        os << "}\n";
                    }
                    for (std::uint32_t r = r0; r < r1; ++r) {
                        const auto set = step_set(i, r);
                        os << "{\n";
                        if (nopp) {
                            lds_step(i, r, 0u);
                        } else {
                            if (r + 1u < r1) {
                                lds_step(i, r + 1u, 1u - set);
                            } else if (i + 1u < T) {
                                lds_step(i + 1u, r0, 1u - set);
                            }
                            os << "__builtin_amdgcn_sched_barrier(0);\n";
                        }
                        for (std::uint32_t c = 0; c < 3u; ++c) {
                            os << "const double pi" << c << " = " << raw(set, 0, c, 0) << " - " << raw(set, 0, c, 1) << ";\n";
                            os << "const double pp" << c << " = " << raw(set, 1, c, 0) << " - " << raw(set, 1, c, 1) << ";\n";
                        }
                        const auto ai = i < M ? nm2("ca", i, r) : gname("al", i, r);
                        const auto bi = i < M ? nm2("cb", i, r) : gname("bl", i, r);
                        auto aph = gname("ap", i, r), bph = gname("bp", i, r);
                        if (xpf && i >= 2u && i <= D + 1u) {
                            aph = xname("xa", i, r - r0);
                            bph = xname("xb", i, r - r0);
                        }
                        if (i >= 2u && i <= hand) {
                            aph = "ah" + S(i) + "_" + S(r);
                            bph = "bh" + S(i) + "_" + S(r);
                        }
                        if (reg_high && !reg_high_br && i >= 2u && i < M) {
                            // (k - i = m < M: the register copy of row m.)
                            for (std::uint32_t m = i; m < M; ++m) {
                                aph = "(k == " + S(i + m) + "u ? " + nm2("ca", m, r) + " : " + aph + ")";
                                bph = "(k == " + S(i + m) + "u ? " + nm2("cb", m, r) + " : " + bph + ")";
                            }
                            os << "const double aph = " << aph << ", bph = " << bph << ";\n";
                            aph = "aph";
                            bph = "bph";
                        }
                        os << "const double bpw = " << bph << " * w2;\n";
                        os << "double sqt = pi0 * pp0;\nsqt = __builtin_fma(pi1, pp1, sqt);\nsqt = __builtin_fma(pi2, pp2, sqt);\n";
                        os << "ssq_" << r << " = __builtin_fma(wq, sqt, ssq_" << r << ");\n";
                        os << "spw_" << r << " = __builtin_fma(ci, " << aph << " * " << bi << ", spw_" << r << ");\n";
                        os << "spw_" << r << " = __builtin_fma(cp, " << ai << " * bpw, spw_" << r << ");\n";
                        for (std::uint32_t c = 0; c < 3u; ++c) {
                            os << "sf" << c << "_" << r << " = __builtin_fma(pp" << c << ", " << bi << ", sf" << c << "_" << r
                               << ");\n";
                            os << "sf" << c << "_" << r << " = __builtin_fma(pi" << c << ", bpw, sf" << c << "_" << r << ");\n";
                        }
                        os << "}\n";
                    }
                    if (i + 1u < T) {
                        os << "if (nm1 == " << i << "u) goto hy_done_" << gi << ";\n";
                    }
                    os << "}\n";
                }
                os << "}\nhy_done_" << gi << ":;\n";
                if (xpf) {
                    const bool last = r1 >= R;
                    const auto n0 = last ? 0u : r1;
                    // (Next group of this order, or the first group of the next order: rows k' - min(floor(k' / 2), i).)
                    os << (last ? "if (k + 1u < " + S(P) + "u) {\nconst unsigned kn = k + 1u, tan = tpa + " + U(rowb) + ", tbn = tpb + " + U(rowb) + ";\n"
                                : std::string("{\nconst unsigned kn = k, tan = tpa, tbn = tpb;\n"));
                    os << "const unsigned nmn = kn >> 1;\n";
                    for (std::uint32_t q = 0; q < G; ++q) {
                        os << "unsigned xcl" << q << " = " << cl_expr(n0 + q) << ";\nasm volatile(\"\" : \"+v\"(xcl" << q << "));\n";
                        os << "const unsigned xlo" << q << " = xcl" << q << " * 8u;\n";
                    }
                    for (std::uint32_t i = 2; i <= D + 1u; ++i) {
                        os << "{\nconst unsigned bk_ = (nmn < " << i << "u ? nmn : " << i << "u) * " << U(rowb) << ";\n";
                        for (std::uint32_t q = 0; q < G; ++q) {
                            os << xname("xa", i, q) << " = HY_TLD(xlo" << q << ", tan - bk_);\n";
                            os << xname("xb", i, q) << " = HY_TLD(xlo" << q << ", tbn - bk_);\n";
                        }
                        os << "}\n";
                    }
                    os << "}\n";
                }
            }
            // Slot 0, second half, and the quotient of the pow recurrence (src/math/pow.cpp:546-549); stores; the register
            // copies of the coefficients of order < M.
            for (std::uint32_t r = r0; r < r1; ++r) {
                os << "{\n";
                unpack_ex(r);
                os << "const unsigned dvo0 = " << out_expr(0, r) << ", dvo1 = " << out_expr(1, r) << ", dvo2 = " << out_expr(2, r)
                   << ";\n";
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    if (keepdz) {
                        os << "const double dz" << c << " = hdz" << c << "_" << r << ";\n";
                    } else {
                        diff("dz" + S(c), c, U(ej0b));
                    }
                }
                os << "const double ak = ssq_" << r << " + ssq_" << r << ";\n";
                os << "const double spw = __builtin_fma(c0k, ak * " << nm2("cb", 0, r) << ", spw_" << r << ");\n";
                if (recip) {
                    // (The quotient of the recurrence as a product with RN(1 / r2^[0]) RN(1 / k) and one correction with the
                    // exact residual: 5 operations instead of the ~13 of a division, correctly rounded but for rare ties.)
                    os << "const double rinv = rca_" << r << " * rkd, den = kd * " << nm2("ca", 0, r) << ";\n";
                    os << "const double bq = spw * rinv;\n";
                    os << "const double bk = __builtin_fma(__builtin_fma(-bq, den, spw), rinv, bq);\n";
                } else {
                    os << "const double bk = spw / (kd * " << nm2("ca", 0, r) << ");\n";
                }
                for (std::uint32_t c = 0; c < 3u; ++c) {
                    os << "const double sf" << c << " = __builtin_fma(dz" << c << ", bk, sf" << c << "_" << r << ");\n";
                }
                if (!is_full(r)) {
                    os << "if (live_" << r << ") {\n";
                }
                // (Row k is read from the tape by a slot i > hand of an order k + i <= P - 1 with i <= k: rows beyond P - 2 - hand
                // are only ever handed over in registers; a row below min(M, hand + 1) is never read from the tape either - as a
                // low member it is in registers, as a high member it belongs to a slot i <= hand.)
                os << (reg_high ? "if (k >= " + S(M) + "u && k + 2u < " + S(P) + "u) {\n" : "if (k >= " + S(std::min(M, hand + 1u)) + "u && k + " + S(hand + 1u) + "u < " + S(P) + "u) {\n");
                if (il_tape) {
                    os << "HY_TST2(ak, bk, 2u * lo_" << r << ", tpa);\n";
                } else {
                    os << "HY_TST(ak, lo_" << r << ", tpa);\nHY_TST(bk, lo_" << r << ", tpb);\n";
                }
                os << "}\n";
                const std::string v[3] = {"sf0", "sf1", "sf2"};
                store_out(r, v);
                if (!is_full(r)) {
                    os << "}\n";
                }
                for (std::uint32_t h = hand; h >= 3u; --h) {
                    os << "ah" << h << "_" << r << " = ah" << h - 1u << "_" << r << ";\nbh" << h << "_" << r << " = bh" << h - 1u << "_" << r
                       << ";\n";
                }
                if (hand2) {
                    os << "ah2_" << r << " = ah1s_" << r << ";\nbh2_" << r << " = bh1s_" << r << ";\n";
                }
                os << "ap1_" << r << " = ak;\nbp1_" << r << " = bk;\n";
                if (M > 1u) {
                    os << "if (k < " << M << "u) {\n";
                    for (std::uint32_t m = 1; m < M; ++m) {
                        os << nm2("ca", m, r) << " = (k == " << m << "u) ? ak : " << nm2("ca", m, r) << ";\n";
                        os << nm2("cb", m, r) << " = (k == " << m << "u) ? bk : " << nm2("cb", m, r) << ";\n";
                    }
                    os << "}\n";
                }
                os << "}\n";
            }
            os << "}\n";
        };
        for (std::uint32_t r0 = 0, gi = 0; r0 < R; r0 += G, ++gi) {
            emit_group(r0, std::min(R, r0 + G), gi);
        }
        glue_levels(1);
        recursion(true);
        os << "}\n";
        return true;
    };
    if (v2) {
        v2 = emit_pair_rounds_body();
    }

    if (!v2) {
    os << "double m0 = 0.0, mo = 0.0, mom1 = 0.0;\n";
    // Order 0: the state is already in slab[0 .. n_eq) and sjet[0 .. n_eq) (see the module text).
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        os << "{ const unsigned i = tid + " << r * bs << "u; if (i < " << n_eq
           << "u) m0 = hy_max(m0, fabs(slab[i])); }\n";
    }
    for (std::uint32_t k = 0; k < order; ++k) {
        os << "// ---- order " << k << " ----\n";
        for (std::uint32_t lev = 1; lev <= pl.max_level; ++lev) {
            if (lev == pl.cluster_level) {
                emit_cluster(k);
            }
            for (std::size_t g = 0; g < pl.groups.size(); ++g) {
                if (pl.groups[g].level == lev) {
                    emit_glue_group(g, k);
                }
            }
            sync();
        }
        // State-variable recursion (src/taylor_02.cpp:245-287): read rhs^[k], sync, write x^[k+1].
        os << "{\n";
        for (std::uint32_t r = 0; r < sv_rounds; ++r) {
            os << "double xn" << r << " = 0.0;\n";
            os << "{ const unsigned i = tid + " << r * bs << "u; if (i < " << n_eq << "u) {\n";
            os << "const unsigned kd = hy_sv_kind[i];\n";
            os << "if (kd == 0u) xn" << r << " = slab[hy_sv_idx[i]] / " << fp_literal(static_cast<double>(k + 1u))
               << ";\n";
            if (k == 0u) {
                os << "else if (kd == 1u) xn" << r << " = hy_sv_val[i];\n";
                os << "else xn" << r << " = a.pars[(u64)hy_sv_idx[i] * N + s];\n";
            }
            os << "} }\n";
        }
        sync();
        for (std::uint32_t r = 0; r < sv_rounds; ++r) {
            os << "{ const unsigned i = tid + " << r * bs << "u; if (i < " << n_eq << "u) {\n";
            os << "slab[i] = xn" << r << ";\n";
            os << "sjet[" << static_cast<std::uint64_t>(k + 1u) * n_eq << "u + i] = xn" << r << ";\n";
            if (k + 1u == order) {
                os << "mo = hy_max(mo, fabs(xn" << r << "));\n";
            } else if (k + 2u == order) {
                os << "mom1 = hy_max(mom1, fabs(xn" << r << "));\n";
            }
            os << "} }\n";
        }
        sync();
        os << "}\n";
    }
    }
    const auto body = os.str();
    os.str("");
    os.clear();

    // Scratch per workgroup: tape + state jets.
    const std::uint64_t tape_doubles = static_cast<std::uint64_t>(n_tape) * order * ncp;
    const std::uint64_t sjet_doubles = (static_cast<std::uint64_t>(order) + 1u) * n_eq;
    const std::uint64_t per_block = (tape_doubles + sjet_doubles + 63u) / 64u * 64u;
    const std::uint32_t wpb = bs / 64u;

    // v2: the order-0 row of the input jets is recorded together with the state (load / update).
    const std::string ej_rec0
        = v2 ? "{ const unsigned ee = hy_sv_ej[i]; if (ee != 0xffffu) HY_EJW("
                   + std::to_string(static_cast<std::uint64_t>(order - 1u) * n_ej * 8u) + "u + ee * 8u) = x; } "
             : std::string{};
    // ===================== module text =====================
    std::ostringstream src;
    src << prelude;
    emit_detail::emit_dout(src, p, opts);
    src << tbl.str();
    src << "extern \"C\" __global__ void __launch_bounds__(" << bs << ") hy_taylor(const hy_kargs a)\n{\n";
    // (slab[n_slots]: the slot idle lanes write to; slab[n_slots + 1]: a constant 0.0 - the missing operands of the merged
    // sums of the v2 glue.)
    src << "__shared__ double slab[" << n_slots + 2u + (v2_dbuf ? n_eq : 0u) << "];\n";
    if (n_ej != 0u) {
        src << "__shared__ double ejet[" << static_cast<std::uint64_t>(n_ej) * (v2 ? order : order - 1u) << "];\n";
    }
    src << "__shared__ double red[3 * " << wpb << "];\n__shared__ u64 sh_base;\n__shared__ int sh_nfi;\n";
    src << "const unsigned tid = threadIdx.x;\nconst u64 N = a.N;\n";
    // Per-lane buffer of the updated state values (lane = state variable, strided).
    src << "double c_new[" << sv_rounds << "];\n";
    src << "double *const tape = a.scratch + (u64)blockIdx.x * " << per_block << "ull;\n";
    src << "double *const sjet = tape + " << tape_doubles << "ull;\n";
    src << v2_decl;
    src << "if (tid == 0u) slab[" << n_slots + 1u << "] = 0.0;\n";
    src << R"HIP(
for (;;) {
// Pull the next system from the device-side work queue.
__syncthreads();
if (tid == 0u) sh_base = atomicAdd((u64 *)(a.counters + 2), (u64)1);
__syncthreads();
const u64 s = sh_base;
if (s >= N) break;
double t_hi = a.time_hi[s], t_lo = a.time_lo[s];
)HIP";
    for (std::uint32_t i = 0; i < p.n_par; ++i) {
        src << "const double par_" << i << " = a.pars[(u64)" << i << "u * N + s];\n";
    }
    src << "for (unsigned i = tid; i < " << n_eq
        << "u; i += " << bs << "u) { const double x = a.state[(u64)i * N + s]; slab[i] = x; sjet[i] = x; " << ej_rec0
        << "}\n";
    src << R"HIP(
__syncthreads();
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
    src << body;
    // Infinity norms: wave shuffles, then one exchange through LDS.
    for (std::uint32_t m = 1; m < 64u; m *= 2u) {
        src << "m0 = hy_max(m0, __shfl_xor(m0, " << m << ", 64));\n";
        src << "mo = hy_max(mo, __shfl_xor(mo, " << m << ", 64));\n";
        src << "mom1 = hy_max(mom1, __shfl_xor(mom1, " << m << ", 64));\n";
    }
    src << "if ((tid & 63u) == 0u) { red[tid >> 6] = m0; red[" << wpb << "u + (tid >> 6)] = mo; red[" << 2u * wpb
        << "u + (tid >> 6)] = mom1; }\n__syncthreads();\n";
    src << "m0 = red[0]; mo = red[" << wpb << "]; mom1 = red[" << 2u * wpb << "];\n";
    for (std::uint32_t w = 1; w < wpb; ++w) {
        src << "m0 = hy_max(m0, red[" << w << "]); mo = hy_max(mo, red[" << wpb + w << "]); mom1 = hy_max(mom1, red["
            << 2u * wpb + w << "]);\n";
    }
    src << "const double num_rho = (m0 <= 1.0) ? 1.0 : m0;\n";
    src << "const double rho_o = hy_root(num_rho / mo, " << fp_literal(1. / static_cast<double>(order)) << ");\n";
    src << "const double rho_om1 = hy_root(num_rho / mom1, " << fp_literal(1. / static_cast<double>(order - 1u))
        << ");\n";
    src << "const double rho_m = hy_min(rho_o, rho_om1);\n";
    src << "double h = rho_m * " << fp_literal(rhofac(order)) << ";\n";
    src << "h = hy_min(h, fabs(lim));\nh = (lim < 0.0) ? -h : h;\n";
    // Taylor coefficients on request, then the state update.
    src << "if (a.tc != nullptr) {\nfor (unsigned i = tid; i < " << n_eq << "u; i += " << bs
        << "u) for (unsigned k = 0; k <= " << order << "u; ++k) a.tc[((u64)i * " << (order + 1u)
        << "u + k) * N + s] = sjet[k * " << n_eq << "u + i];\n}\n";
    src << "int nfi = 0;\n";
    // NOTE: the coefficients of a state variable are read back from the per-workgroup array in global memory. As a rolled
    // loop this was ONE load per iteration with a full wait behind it (20 dependent round trips to L2 per state variable and
    // step: ~30 us of a 260 us step on nbody(64)); written out, all the loads of all the rounds of the lane are issued
    // before the first operation - the registers of the order loop are free by now. Branch-free: a lane without a state
    // variable in its last round evaluates a copy of the last variable and the result is ignored.
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        const bool partial = static_cast<std::uint64_t>(r + 1u) * bs > n_eq;
        // (Opaque to the optimiser: as invariants of the step loop the order + 1 addresses of a round are computed once per
        // kernel and spilled - 42 registers reloaded from scratch at every step.)
        src << "unsigned hi" << r << " = " << (partial ? "(tid + " + std::to_string(r * bs) + "u < " + std::to_string(n_eq)
                                                                + "u) ? tid + " + std::to_string(r * bs) + "u : "
                                                                + std::to_string(n_eq - 1u) + "u"
                                                          : "tid + " + std::to_string(r * bs) + "u")
            << ";\nasm volatile(\"\" : \"+v\"(hi" << r << "));\n";
        for (std::uint32_t k = 0; k <= order; ++k) {
            src << "const double hc" << r << "_" << k << " = sjet[" << static_cast<std::uint64_t>(k) * n_eq << "u + hi" << r
                << "];\n";
        }
    }
    for (std::uint32_t r = 0; r < sv_rounds; ++r) {
        const auto c = [&](std::uint32_t k) { return "hc" + std::to_string(r) + "_" + std::to_string(k); };
        src << "{\n";
        if (opts.high_accuracy) {
            src << "double res = " << c(0) << ", comp = 0.0, cur_h = h;\n";
            for (std::uint32_t k = 1; k <= order; ++k) {
                src << "{\nconst double tmp = " << c(k) << " * cur_h;\nconst double y = tmp - comp;\n";
                src << "const double t = res + y;\ncomp = (t - res) - y;\nres = t;\ncur_h = cur_h * h;\n}\n";
            }
        } else {
            src << "double res = " << c(order) << ";\n";
            for (std::uint32_t k = 1; k <= order; ++k) {
                src << "res = " << c(order - k) << " + res * h;\n";
            }
        }
        src << "if (!hy_finite(res)) nfi = 1;\n";
        // NOTE: sjet row 0 / slab are rewritten only after every lane is done reading the jets.
        src << "c_new[" << r << "] = res;\n}\n";
    }
    src << R"HIP(
{
    hy_df tcur; tcur.hi = t_hi; tcur.lo = t_lo;
    hy_df hh; hh.hi = h; hh.lo = 0.0;
    const hy_df nt = hy_df_add(tcur, hh);
    t_hi = nt.hi; t_lo = nt.lo;
}
last_h = h;
if (!(hy_finite(t_hi) && hy_finite(t_lo))) nfi = 1;
nfi = __syncthreads_or(nfi);
)HIP";
    src << "for (unsigned i = tid; i < " << n_eq << "u; i += " << bs << "u) { const double x = c_new[i / " << bs
        << "u]; slab[i] = x; sjet[i] = x; " << ej_rec0 << "}\n__syncthreads();\n";
    src << R"HIP(
HY_STEP_TAIL(nfi != 0, tid == 0u)
}
)HIP";
    src << "for (unsigned i = tid; i < " << n_eq << "u; i += " << bs << "u) a.state[(u64)i * N + s] = slab[i];\n";
    src << R"HIP(
if (tid == 0u) {
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

    ret.source = src.str();
    ret.kernel_name = "hy_taylor";
    ret.dout_name = "hy_dout";
    ret.block_size = bs;
    ret.lanes_per_system = bs;
    ret.n_clusters = nc;
    ret.lds_bytes = 0;
    ret.mode = emit_mode::block;
    ret.n_statements = e.n_stmt;
    ret.scratch_per_wave = per_block / wpb;
    ret.persistent = true;
    ret.tc_optional = true;
    ret.notes = "block mode: one system per workgroup of " + std::to_string(bs) + " lanes, " + std::to_string(nc)
                + " clusters of " + std::to_string(t0.size()) + " nodes (" + std::to_string(n_tape)
                + " members on the tape, " + std::to_string(n_sto - n_tape) + " recomputed from " + std::to_string(n_ej)
                + " LDS-resident input jets), " + std::to_string(pl.groups.size()) + " glue groups, "
                + std::to_string(n_slots) + " LDS slots, tape " + std::to_string(per_block * 8u / 1024u)
                + " KiB per workgroup";
    // Machine-level loop-invariant code motion off for the v2 kernel: invariants of the step loop hoisted in front of it
    // are what the register allocator spills (nbody(64): 54 -> 41 spilled registers, +1.3 %; "licm": A/B harness).
    if (v2 && ("," + opts.dev.block_opts + ",").find(",licm,") == std::string::npos) {
        ret.compile_flags = "-mllvm -disable-machine-licm";
    }
    if (v2) {
        ret.notes += "; v2 cluster phase: rolled order loop, " + std::to_string(n_iter) + " rounds per lane, rows < "
                     + std::to_string(v2_M) + " of the tape members in registers, the " + std::to_string(v2_hand)
                     + " most recent rows handed over in registers";
    }
    return ret;
}

} // namespace heyoka_amd
