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
#include <string>
#include <vector>

#include "hip_emit_cluster_plan.hpp"
#include "hip_emit_detail.hpp"

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
    why_not = make_plan(p, opts.order, pl, lim);
    if (!why_not.empty()) {
        return ret;
    }

    const auto n_eq = p.n_eq, order = opts.order;
    // Workgroup size: 256 lanes = one wavefront per SIMD with the whole register file (512 VGPRs) for the
    // history of the cluster being processed.
    std::uint32_t bs = 256;
    {
        // Small cluster counts: do not park lanes (several smaller workgroups fit on a compute unit).
        const auto nc0 = static_cast<std::uint32_t>(pl.clusters.size());
        if (nc0 <= 128u) {
            bs = 128;
        }
    }
    if (const char *ev = std::getenv("HEYOKA_AMD_BLOCK_SIZE")) {
        const auto v = std::atoi(ev);
        if (v == 128 || v == 256 || v == 512 || v == 1024) {
            bs = static_cast<std::uint32_t>(v);
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
    // streams of the N-body clusters (HEYOKA_AMD_BLOCK_NO_RECOMPUTE=1 disables it, for A/B measurements).
    std::vector<char> recomp(n_sto, 0);
    std::vector<char> ext_used(n_ext, 0);
    std::vector<std::uint32_t> ej_slots; // distinct slab slots of the external inputs with LDS jets
    std::vector<std::uint32_t> ej_index; // [x * nc + c] -> index into ej_slots
    if (std::getenv("HEYOKA_AMD_BLOCK_NO_RECOMPUTE") == nullptr) {
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
        std::map<std::uint32_t, std::uint32_t> slot_to_ej;
        ej_index.assign(static_cast<std::size_t>(n_ext) * nc, 0u);
        for (std::uint32_t x = 0; x < n_ext; ++x) {
            if (ext_used[x] == 0) {
                continue;
            }
            for (std::uint32_t c = 0; c < nc; ++c) {
                const auto slot = static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[c][x]]);
                auto it = slot_to_ej.find(slot);
                if (it == slot_to_ej.end()) {
                    it = slot_to_ej.emplace(slot, static_cast<std::uint32_t>(ej_slots.size())).first;
                    ej_slots.push_back(slot);
                }
                ej_index[static_cast<std::size_t>(x) * nc + c] = it->second;
            }
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
    const auto emit_utbl = [&](const std::string &name, const std::vector<std::uint32_t> &v) {
        tbl << "__device__ const " << ut << " " << name << "[" << std::max<std::size_t>(v.size(), 1u) << "] = {";
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
        if (const char *ev = std::getenv("HEYOKA_AMD_BLOCK_PREFETCH")) {
            return std::max(0, std::min(2, std::atoi(ev)));
        }
        return 0;
    }();
    const std::uint32_t sched_every = [&]() -> std::uint32_t {
        if (const char *ev = std::getenv("HEYOKA_AMD_BLOCK_SCHED")) {
            return static_cast<std::uint32_t>(std::max(1, std::atoi(ev)));
        }
        return 4u;
    }();
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
    const auto body = os.str();
    os.str("");
    os.clear();

    // Scratch per workgroup: tape + state jets.
    const std::uint64_t tape_doubles = static_cast<std::uint64_t>(n_tape) * order * ncp;
    const std::uint64_t sjet_doubles = (static_cast<std::uint64_t>(order) + 1u) * n_eq;
    const std::uint64_t per_block = (tape_doubles + sjet_doubles + 63u) / 64u * 64u;
    const std::uint32_t wpb = bs / 64u;

    // ===================== module text =====================
    std::ostringstream src;
    src << prelude;
    emit_detail::emit_dout(src, p, opts);
    src << tbl.str();
    src << "extern \"C\" __global__ void __launch_bounds__(" << bs << ") hy_taylor(const hy_kargs a)\n{\n";
    src << "__shared__ double slab[" << n_slots + 1u << "];\n";
    if (n_ej != 0u) {
        src << "__shared__ double ejet[" << static_cast<std::uint64_t>(n_ej) * (order - 1u) << "];\n";
    }
    src << "__shared__ double red[3 * " << wpb << "];\n__shared__ u64 sh_base;\n__shared__ int sh_nfi;\n";
    src << "const unsigned tid = threadIdx.x;\nconst u64 N = a.N;\n";
    // Per-lane buffer of the updated state values (lane = state variable, strided).
    src << "double c_new[" << sv_rounds << "];\n";
    src << "double *const tape = a.scratch + (u64)blockIdx.x * " << per_block << "ull;\n";
    src << "double *const sjet = tape + " << tape_doubles << "ull;\n";
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
        << "u; i += " << bs << "u) { const double x = a.state[(u64)i * N + s]; slab[i] = x; sjet[i] = x; }\n";
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
    src << "for (unsigned i = tid; i < " << n_eq << "u; i += " << bs << "u) {\nconst double *c = sjet + i;\n";
    if (opts.high_accuracy) {
        src << "double res = c[0], comp = 0.0, cur_h = h;\n";
        src << "for (unsigned k = 1; k <= " << order << "u; ++k) {\n";
        src << "const double tmp = c[k * " << n_eq << "u] * cur_h;\nconst double y = tmp - comp;\n";
        src << "const double t = res + y;\ncomp = (t - res) - y;\nres = t;\ncur_h = cur_h * h;\n}\n";
    } else {
        src << "double res = c[" << static_cast<std::uint64_t>(order) * n_eq << "u];\n";
        src << "for (unsigned k = 1; k <= " << order << "u; ++k) {\n";
        src << "res = c[(" << order << "u - k) * " << n_eq << "u] + res * h;\n}\n";
    }
    src << "if (!hy_finite(res)) nfi = 1;\n";
    // NOTE: sjet row 0 / slab are rewritten only after every lane is done reading the jets.
    src << "c_new[i / " << bs << "u] = res;\n}\n";
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
        << "u]; slab[i] = x; sjet[i] = x; }\n__syncthreads();\n";
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
    return ret;
}

} // namespace heyoka_amd
