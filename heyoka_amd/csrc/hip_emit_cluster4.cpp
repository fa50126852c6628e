// Cluster code generation, version 4: wave roles.
//
// Point-mass pair clusters (cluster_detail::pair_pattern) are split over TWO WAVEFRONTS of a 128-thread workgroup
// instead of two lanes of one wavefront (v3, hip_emit_cluster2.cpp):
//   role A (wavefront 0): the coordinate differences d_0, d_1, their squares and their products with sa = c * pow;
//   role B (wavefront 1): d_2, the sum of squares b, the pow recurrence and d_2 * sa.
// Lane l of a 16-lane group = pair l of one of the 4 systems of the workgroup, in both wavefronts. Each role is its own
// straight-line instruction stream (a wave-uniform branch on the wavefront index): no lane executes work of the other
// role (v3 runs the union of both on every lane), the per-system work which is not convolutions - glue sums, state
// recursion, final evaluation - is shared between the two wavefronts instead of replicated per 2 systems, and a lane
// keeps 3 coefficient histories (120 registers): 256 registers, 4 workgroups = 8 wavefronts per CU.
//
// The two roles exchange through LDS and meet at workgroup barriers ("ticks"). Order k takes four phases
//   P1(k) tick 2k    : A, B: differences of order k, squares; A publishes its partial sum of squares
//   P2(k) tick 2k + 1: B: b_k, pow recurrence -> sa_k (published), d_2 * sa
//   P3(k) tick 2k + 2: A: d_0 * sa, d_1 * sa
//   P4(k) tick 2k + 3: A, B: glue(k) (accelerations of order k) -> v^[k+1], x^[k+2]
// and, because a position of order k + 2 only needs the accelerations of order k, the even and the odd orders are
// interleaved: tick 2k carries P1(k) and P3(k-1), tick 2k + 1 carries P2(k) and P4(k-1) - two barriers per order.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>

#include "hip_emit_cluster_plan.hpp"
#include "hip_emit_detail.hpp"

namespace heyoka_amd
{

emitted_module emit_cluster_v4(const taylor_program &p, const emit_options &opts, std::string &why_not)
{
    using cluster_detail::cluster_plan;
    using cluster_detail::is_var;
    using emit_detail::prelude;
    using emit_detail::rhofac;
    using emit_detail::ssa_emitter;

    emitted_module ret;
    cluster_plan pl;
    why_not = cluster_detail::make_plan(p, opts.order, pl);
    if (!why_not.empty()) {
        return ret;
    }
    const auto n_eq = p.n_eq, order = opts.order;
    const auto nc = static_cast<std::uint32_t>(pl.clusters.size());
    cluster_detail::pair_pattern pp;
    cluster_detail::detect_pair_pattern(p, pl, pp);
    if (!pp.ok || nc > 16u || p.n_par != 0u || pl.cluster_level != 1u || pl.max_level != 2u || order < 3u) {
        why_not = "wave-role kernel: not a point-mass pair system with at most 16 pairs";
        return ret;
    }
    for (const auto &g : pl.groups) {
        if (g.level != 2u) {
            why_not = "wave-role kernel: glue below the clusters";
            return ret;
        }
    }
    const std::uint32_t L = 16, spw = 4, bs = 128;
    const bool has_rx = pp.rx[0] >= 0;

    // ---- 1. State-variable chains anchored at glue nodes (as in v2). ----
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
    std::map<std::uint32_t, std::vector<std::uint32_t>> att;
    for (std::uint32_t i = 0; i < n_eq; ++i) {
        att[static_cast<std::uint32_t>(anchor[i])].push_back(i);
    }
    for (auto &[u, v] : att) {
        std::sort(v.begin(), v.end(), [&](std::uint32_t a, std::uint32_t b) { return depth[a] < depth[b]; });
        for (std::size_t j = 0; j < v.size(); ++j) {
            const auto &def = p.sv_defs[v[j]];
            if (depth[v[j]] != j + 1u || def.idx != ((j == 0u) ? u : v[j - 1u])) {
                why_not = "branching state-variable chains";
                return ret;
            }
        }
    }
    // One glue group of identical shape, every node with the same number of attached variables.
    if (pl.groups.size() != 1u) {
        why_not = "wave-role kernel: more than one glue group";
        return ret;
    }
    const auto &grp = pl.groups[0];
    const auto n_nodes = static_cast<std::uint32_t>(grp.nodes.size());
    const auto natt = [&](std::uint32_t u) {
        const auto it = att.find(u);
        return it == att.end() ? 0u : static_cast<std::uint32_t>(it->second.size());
    };
    const auto n_att = natt(grp.nodes[0]);
    for (const auto u : grp.nodes) {
        if (natt(u) != n_att) {
            why_not = "glue nodes with different state-variable chains";
            return ret;
        }
    }
    if (n_att == 0u || n_nodes > 2u * L || n_nodes * n_att != n_eq) {
        why_not = "wave-role kernel: unsupported glue / state-variable structure";
        return ret;
    }
    // Nodes of the two roles.
    const auto n_a = (n_nodes + 1u) / 2u, n_b = n_nodes - n_a;
    const std::uint32_t role_first[2] = {0u, n_a}, role_count[2] = {n_a, n_b};

    // ---- 2. LDS slots (per system): state variables, cluster outputs in lane order, glue nodes, exchange slots. ----
    std::fill(pl.slot_of.begin(), pl.slot_of.end(), -1);
    std::uint32_t ns = 0;
    // Jet columns / state slots: role, chain position, lane.
    std::vector<std::uint32_t> col_of(n_eq, 0);
    std::uint32_t n_col = 0;
    for (std::uint32_t r = 0; r < 2u; ++r) {
        for (std::uint32_t a = 0; a < n_att; ++a) {
            for (std::uint32_t j = 0; j < role_count[r]; ++j) {
                const auto var = att.at(grp.nodes[role_first[r] + j])[a];
                pl.slot_of[var] = static_cast<int>(ns++);
                col_of[var] = n_col++;
            }
        }
    }
    for (std::uint32_t c3 = 0; c3 < 3u; ++c3) {
        for (std::uint32_t c = 0; c < nc; ++c) {
            pl.slot_of[pl.clusters[c][pp.pr[c3]]] = static_cast<int>(ns++);
        }
    }
    if (has_rx) {
        for (std::uint32_t c3 = 0; c3 < 3u; ++c3) {
            for (std::uint32_t c = 0; c < nc; ++c) {
                pl.slot_of[pl.clusters[c][static_cast<std::uint32_t>(pp.rx[c3])]] = static_cast<int>(ns++);
            }
        }
    }
    const auto n_par_slots = ns;           // double-buffered by order parity
    const auto dummy_base = ns;            // 4 dummy slots (idle lanes)
    ns += 4u;
    const auto buf_stride = ns;
    const auto xch_a = 2u * ns;            // single-buffered exchange slots: partial sums of squares (A -> B)
    const auto xch_b = xch_a + L;          // sa_k (B -> A)
    const auto slab_stride = (xch_b + L) | 1u;
    (void)n_par_slots;
    const auto n_colp = n_col + 1u;
    const auto n_hslots = (n_col + L - 1u) / L;
    const auto jet_doubles = static_cast<std::uint64_t>(order + 1u) * spw * n_colp;
    if ((static_cast<std::uint64_t>(spw) * slab_stride + jet_doubles + 64u) * 8u > 40u * 1024u) {
        why_not = "wave-role kernel: the workgroup's slab + jets exceed 40 KB of LDS";
        return ret;
    }

    // ---- 3. Lane tables. ----
    std::vector<std::vector<std::uint32_t>> utbl;
    std::vector<std::vector<double>> dtbl;
    const auto add_utbl = [&](std::vector<std::uint32_t> v) {
        for (std::size_t t = 0; t < utbl.size(); ++t) {
            if (utbl[t] == v) {
                return t;
            }
        }
        utbl.push_back(std::move(v));
        return utbl.size() - 1u;
    };
    const auto add_dtbl = [&](std::vector<double> v) {
        dtbl.push_back(std::move(v));
        return dtbl.size() - 1u;
    };
    const auto utname = [](std::size_t t) { return "ut" + std::to_string(t); };
    const auto dtname = [](std::size_t t) { return "dt" + std::to_string(t); };
    const auto lane_cluster = [&](std::uint32_t l) { return l < nc ? l : 0u; };
    const auto ext_slot = [&](std::uint32_t l, std::uint32_t dc, std::uint32_t a) {
        return static_cast<std::uint32_t>(pl.slot_of[pl.ext_u[lane_cluster(l)][pp.de[dc][a]]]);
    };
    const auto member_slot = [&](std::uint32_t l, std::uint32_t q, std::uint32_t dflt) {
        return l < nc ? static_cast<std::uint32_t>(pl.slot_of[pl.clusters[l][q]]) : dflt;
    };
    const auto lane_tbl = [&](const auto &f) {
        std::vector<std::uint32_t> v(L);
        for (std::uint32_t l = 0; l < L; ++l) {
            v[l] = f(l);
        }
        return add_utbl(std::move(v));
    };
    const auto lane_dtbl = [&](const auto &f) {
        std::vector<double> v(L);
        for (std::uint32_t l = 0; l < L; ++l) {
            v[l] = f(l);
        }
        return add_dtbl(std::move(v));
    };
    std::size_t t_ext[3][2], t_pr[3], t_rx[3] = {0, 0, 0}, t_crx[3] = {0, 0, 0}, t_csc = 0;
    for (std::uint32_t c3 = 0; c3 < 3u; ++c3) {
        for (std::uint32_t a = 0; a < 2u; ++a) {
            t_ext[c3][a] = lane_tbl([&](std::uint32_t l) { return ext_slot(l, c3, a); });
        }
        t_pr[c3] = lane_tbl([&](std::uint32_t l) { return member_slot(l, pp.pr[c3], dummy_base + c3); });
        if (has_rx) {
            t_rx[c3] = lane_tbl([&](std::uint32_t l) {
                return member_slot(l, static_cast<std::uint32_t>(pp.rx[c3]), dummy_base + 3u);
            });
            t_crx[c3] = lane_dtbl([&](std::uint32_t l) {
                return p.nodes[pl.clusters[lane_cluster(l)][static_cast<std::uint32_t>(pp.rx[c3])] - n_eq].args[0].value;
            });
        }
    }
    if (pp.sc >= 0) {
        t_csc = lane_dtbl([&](std::uint32_t l) {
            return p.nodes[pl.clusters[lane_cluster(l)][static_cast<std::uint32_t>(pp.sc)] - n_eq].args[0].value;
        });
    }
    // Glue: per role, lane j < role_count handles node role_first + j (idle lanes replicate lane 0).
    struct owner_slot {
        std::size_t out_tbl = 0, var_tbl = 0, col_tbl = 0;
        bool slab_needed = false;
        std::vector<std::string> xname;
    };
    struct role_glue {
        std::vector<std::size_t> arg_tbl;
        std::vector<owner_slot> owners;
    } rg[2];
    std::vector<char> read_thru_slab(p.n_u, 0);
    for (const auto &n : p.nodes) {
        for (const auto &o : n.args) {
            if (is_var(o)) {
                read_thru_slab[o.idx] = 1;
            }
        }
    }
    const auto &n0 = p.nodes[grp.nodes[0] - n_eq];
    for (std::uint32_t r = 0; r < 2u; ++r) {
        const auto node_of = [&](std::uint32_t l) { return grp.nodes[role_first[r] + (l < role_count[r] ? l : 0u)]; };
        for (std::size_t a = 0; a < n0.args.size(); ++a) {
            if (is_var(n0.args[a])) {
                rg[r].arg_tbl.push_back(lane_tbl([&](std::uint32_t l) {
                    return static_cast<std::uint32_t>(pl.slot_of[p.nodes[node_of(l) - n_eq].args[a].idx]);
                }));
            } else if (n0.args[a].type == operand::kind::num) {
                rg[r].arg_tbl.push_back(lane_dtbl([&](std::uint32_t l) { return p.nodes[node_of(l) - n_eq].args[a].value; }));
            } else {
                rg[r].arg_tbl.push_back(0);
            }
        }
        for (std::uint32_t a = 0; a < n_att; ++a) {
            owner_slot ow;
            ow.out_tbl = lane_tbl([&](std::uint32_t l) {
                return l < role_count[r] ? static_cast<std::uint32_t>(pl.slot_of[att.at(node_of(l))[a]]) : dummy_base;
            });
            ow.var_tbl = lane_tbl([&](std::uint32_t l) { return att.at(node_of(l))[a]; });
            // Jet column: the idle lanes write to the dummy column and read the column of the replicated variable.
            ow.col_tbl = lane_tbl([&](std::uint32_t l) { return col_of[att.at(node_of(l))[a]]; });
            for (std::uint32_t l = 0; l < role_count[r]; ++l) {
                ow.slab_needed = ow.slab_needed || read_thru_slab[att.at(node_of(l))[a]] != 0;
            }
            ow.xname.resize(order + 1u);
            rg[r].owners.push_back(std::move(ow));
        }
    }

    // ---- 4. Emission helpers (one emitter per role). ----
    const auto slabk = [&](std::uint32_t k, const std::string &idx) {
        return (k % 2u == 0u) ? ("slab[" + idx + "]") : ("slab[" + idx + " + " + std::to_string(buf_stride) + "u]");
    };
    const auto kstride = static_cast<std::uint64_t>(spw) * n_colp;
    const auto jrow = [&](std::uint32_t k) { return std::to_string(static_cast<std::uint64_t>(k) * kstride); };

    std::string role_body[2];
    for (std::uint32_t r = 0; r < 2u; ++r) {
        ssa_emitter e(p, order);
        auto &os = e.os;
        const char *rn = (r == 0u) ? "A" : "B";
        auto &glue = rg[r];
        const auto mul = [](const std::string &a, const std::string &b) { return a + " * " + b; };

        const auto publish_sv = [&](owner_slot &ow, std::size_t oi, std::uint32_t k, const std::string &name) {
            ow.xname[k] = name;
            if (ow.slab_needed) {
                os << slabk(k, utname(ow.out_tbl)) << " = " << name << ";\n";
            }
            if (k != 0u) {
                os << "jw" << rn << oi << "[" << jrow(k) << "] = " << name << ";\n";
            }
            const char *acc = (k == 0u) ? "m0" : (k == order ? "mo" : (k == order - 1u ? "mom1" : nullptr));
            if (acc != nullptr) {
                os << acc << " = hy_max(" << acc << ", fabs(" << name << "));\n";
            }
        };
        const auto glue_reads = [&](std::uint32_t k) {
            std::vector<std::string> names(n0.args.size());
            for (std::size_t a = 0; a < n0.args.size(); ++a) {
                if (is_var(n0.args[a])) {
                    names[a] = e.def(slabk(k, utname(glue.arg_tbl[a])));
                }
            }
            return names;
        };
        const auto glue_compute = [&](std::uint32_t k, const std::vector<std::string> &names) {
            const auto rep = grp.nodes[0];
            std::vector<std::pair<std::uint32_t, std::string>> saved_vals;
            const auto saved = e.numpar_override;
            for (std::size_t a = 0; a < n0.args.size(); ++a) {
                const auto &o = n0.args[a];
                if (is_var(o)) {
                    saved_vals.emplace_back(o.idx, e.val(o.idx, k));
                    e.val(o.idx, k) = names[a];
                } else if (o.type == operand::kind::num) {
                    e.numpar_override[&o] = dtname(glue.arg_tbl[a]);
                }
            }
            if (n0.kind == func_kind::prod && n0.args[0].type == operand::kind::num && n0.args[0].value == -1.) {
                e.numpar_override.erase(&n0.args[0]);
            }
            e.node(rep - n_eq, k);
            const auto gval = e.val(rep, k);
            for (auto it = saved_vals.rbegin(); it != saved_vals.rend(); ++it) {
                e.val(it->first, k) = it->second;
            }
            e.numpar_override = saved;
            for (std::size_t a = 0; a < glue.owners.size(); ++a) {
                const auto ord = k + 1u + static_cast<std::uint32_t>(a);
                if (ord > order) {
                    continue;
                }
                const auto src = (a == 0u) ? gval : glue.owners[a - 1u].xname[ord - 1u];
                publish_sv(glue.owners[a], a, ord, e.div_const(src, ord));
            }
        };

        // Coefficient histories and history chains of the role.
        std::vector<std::string> d0(order + 1u), d1(order + 1u), sa(order + 1u), bb(order + 1u);
        std::string h_c0, h_c1, h_s0, h_s1, h_m0, h_m1, h_pw, pB, rb1;

        // ---- order 0 of the state variables + the orders which follow from the state alone ----
        os << "double m0 = 0.0, mo = 0.0, mom1 = 0.0;\n";
        for (std::size_t oi = 0; oi < glue.owners.size(); ++oi) {
            os << "const double xs" << rn << oi << " = jr" << rn << oi << "[0];\n";
            publish_sv(glue.owners[oi], oi, 0, std::string("xs") + rn + std::to_string(oi));
        }
        for (std::size_t a = 1; a < glue.owners.size(); ++a) {
            for (std::uint32_t j = 1; j <= a && j <= order; ++j) {
                publish_sv(glue.owners[a], a, j, e.div_const(glue.owners[a - 1u].xname[j - 1u], j));
            }
        }
        os << "__syncthreads();\n";

        const std::uint32_t n_ticks = 2u * order + 2u;
        for (std::uint32_t T = 0; T < n_ticks; ++T) {
            const auto k = T / 2u;
            if (T % 2u == 0u) {
                // ---- even tick: P1(k) and (role A) P3(k - 1) ----
                std::vector<std::string> rd;
                std::string sa_in;
                if (k < order) {
                    const auto rds = [&](std::uint32_t c3) {
                        rd.push_back(e.def(slabk(k, utname(t_ext[c3][0]))));
                        rd.push_back(e.def(slabk(k, utname(t_ext[c3][1]))));
                    };
                    if (r == 0u) {
                        rds(0);
                        rds(1);
                    } else {
                        rds(2);
                    }
                }
                if (r == 0u && k >= 1u && k - 1u < order) {
                    sa_in = e.def("slab[" + std::to_string(xch_b) + "u + l]");
                }
                if (k < order) {
                    // Differences and squares of order k.
                    const auto sq_finish = [&](std::vector<std::string> &d, const std::string &hs, const std::string &hm) {
                        if (k == 0u) {
                            return e.def(mul(d[0], d[0]));
                        }
                        const auto acc = e.chain(hs, d[k], d[0]);
                        return (k % 2u == 0u) ? e.def("__builtin_fma(2.0, " + acc + ", " + hm + ")") : e.def(acc + " + " + acc);
                    };
                    d0[k] = e.def(rd[0] + " - " + rd[1]);
                    const auto s0 = sq_finish(d0, h_s0, h_m0);
                    if (r == 0u) {
                        d1[k] = e.def(rd[2] + " - " + rd[3]);
                        const auto s1 = sq_finish(d1, h_s1, h_m1);
                        const auto pa = e.def(s0 + " + " + s1);
                        os << "slab[" << xch_a << "u + l] = " << pa << ";\n";
                    } else {
                        pB = s0;
                    }
                }
                if (r == 0u && k >= 1u && k - 1u < order) {
                    // P3(k - 1): the products of role A.
                    const auto km = k - 1u;
                    sa[km] = sa_in;
                    std::string p0, p1;
                    if (km == 0u) {
                        p0 = e.def(mul(d0[0], sa[0]));
                        p1 = e.def(mul(d1[0], sa[0]));
                    } else {
                        p0 = e.chain(e.chain(h_c0, d0[km], sa[0]), d0[0], sa[km]);
                        p1 = e.chain(e.chain(h_c1, d1[km], sa[0]), d1[0], sa[km]);
                    }
                    os << slabk(km, utname(t_pr[0])) << " = " << p0 << ";\n";
                    os << slabk(km, utname(t_pr[1])) << " = " << p1 << ";\n";
                    if (has_rx) {
                        const auto r0 = e.def(mul(dtname(t_crx[0]), p0));
                        const auto r1 = e.def(mul(dtname(t_crx[1]), p1));
                        os << slabk(km, utname(t_rx[0])) << " = " << r0 << ";\n";
                        os << slabk(km, utname(t_rx[1])) << " = " << r1 << ";\n";
                    }
                    // History of the products of order k (indices 1 .. k - 1): needed at tick 2 k + 2.
                    h_c0.clear();
                    h_c1.clear();
                    if (k < order) {
                        for (std::uint32_t j = 1; j < k; ++j) {
                            h_c0 = e.chain(h_c0, d0[k - j], sa[j]);
                            h_c1 = e.chain(h_c1, d1[k - j], sa[j]);
                        }
                    }
                }
                if (k < order) {
                    // History of the squares of order k + 1: needed at tick 2 k + 2.
                    const auto K = k + 1u;
                    h_s0.clear();
                    h_s1.clear();
                    h_m0.clear();
                    h_m1.clear();
                    if (K < order && K >= 2u) {
                        const auto jmax = (K % 2u == 1u) ? (K - 1u) / 2u : (K - 2u) / 2u;
                        for (std::uint32_t j = 1; j <= jmax; ++j) {
                            h_s0 = e.chain(h_s0, d0[K - j], d0[j]);
                            if (r == 0u) {
                                h_s1 = e.chain(h_s1, d1[K - j], d1[j]);
                            }
                        }
                        if (K % 2u == 0u) {
                            h_m0 = e.def(mul(d0[K / 2u], d0[K / 2u]));
                            if (r == 0u) {
                                h_m1 = e.def(mul(d1[K / 2u], d1[K / 2u]));
                            }
                        }
                    }
                }
            } else {
                // ---- odd tick: (role B) P2(k) and P4(k - 1) ----
                std::string pa_in;
                std::vector<std::string> gnames;
                const bool do_glue = k >= 1u && k - 1u < order;
                if (r == 1u && k < order) {
                    pa_in = e.def("slab[" + std::to_string(xch_a) + "u + l]");
                }
                if (do_glue) {
                    gnames = glue_reads(k - 1u);
                }
                if (r == 1u && k < order) {
                    bb[k] = e.def(pa_in + " + " + pB);
                    if (k == 0u) {
                        const auto a0 = e.pow_eval(bb[0], pp.ex);
                        sa[0] = pp.sc >= 0 ? e.def(mul(dtname(t_csc), a0)) : a0;
                        rb1 = e.def("1.0 / " + bb[0]);
                    } else {
                        // k b_0 sa_k = sum_{j<k} (k alpha - j (alpha + 1)) b_{k-j} sa_j (src/math/pow.cpp:517-549; linear in a:
                        // run on the scaled sa); the division by k b_0 as in v3 (reciprocal + exact-residual correction).
                        const auto t = e.def(mul(bb[k], sa[0]));
                        const auto c0 = fp_literal(pp.ex * static_cast<double>(k));
                        const auto num = h_pw.empty() ? e.def(mul(c0, t)) : e.def(c0 + " * " + t + " + " + h_pw);
                        const auto dv = e.def(mul(fp_literal(static_cast<double>(k)), bb[0]));
                        const auto rk = (k == 1u) ? rb1 : e.def(mul(rb1, fp_literal(1. / static_cast<double>(k))));
                        const auto q0 = e.def(mul(num, rk));
                        const auto rem = e.def("__builtin_fma(-" + dv + ", " + q0 + ", " + num + ")");
                        sa[k] = e.def("__builtin_fma(" + rem + ", " + rk + ", " + q0 + ")");
                    }
                    os << "slab[" << xch_b << "u + l] = " << sa[k] << ";\n";
                    std::string p2;
                    if (k == 0u) {
                        p2 = e.def(mul(d0[0], sa[0]));
                    } else {
                        p2 = e.chain(e.chain(h_c0, d0[k], sa[0]), d0[0], sa[k]);
                    }
                    os << slabk(k, utname(t_pr[2])) << " = " << p2 << ";\n";
                    if (has_rx) {
                        const auto r2 = e.def(mul(dtname(t_crx[2]), p2));
                        os << slabk(k, utname(t_rx[2])) << " = " << r2 << ";\n";
                    }
                }
                if (do_glue) {
                    glue_compute(k - 1u, gnames);
                }
                if (r == 1u && k < order) {
                    // Histories of order k + 1 (indices 1 .. k): the product and the pow recurrence.
                    const auto K = k + 1u;
                    h_c0.clear();
                    h_pw.clear();
                    if (K < order) {
                        for (std::uint32_t j = 1; j < K; ++j) {
                            h_c0 = e.chain(h_c0, d0[K - j], sa[j]);
                            const double sf = static_cast<double>(K) * pp.ex - static_cast<double>(j) * (pp.ex + 1.);
                            const auto pr = e.def(mul(bb[K - j], sa[j]));
                            h_pw = e.chain(h_pw, fp_literal(sf), pr);
                        }
                    }
                }
            }
            os << "__syncthreads();\n";
        }
        // Norms of the role's variables, reduced over the lanes of the system.
        for (std::uint32_t m = 1; m < L; m *= 2u) {
            os << "m0 = hy_max(m0, __shfl_xor(m0, " << m << ", 64));\n";
            os << "mo = hy_max(mo, __shfl_xor(mo, " << m << ", 64));\n";
            os << "mom1 = hy_max(mom1, __shfl_xor(mom1, " << m << ", 64));\n";
        }
        os << "xm_mine[0] = m0;\nxm_mine[1] = mo;\nxm_mine[2] = mom1;\n";
        role_body[r] = os.str();
        ret.n_statements += e.n_stmt;
    }

    // ===================== module text =====================
    std::ostringstream src;
    src << "#define SPW " << spw << "u\n";
    src << prelude;
    emit_detail::emit_dout(src, p, opts);
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
    src << "};\n__constant__ double hy_dtbl[" << std::max<std::size_t>(dtbl.size(), 1u) * L << "] = {";
    for (const auto &v : dtbl) {
        for (const auto x : v) {
            src << fp_literal(x) << ",";
        }
    }
    src << "};\n";
    src << "extern \"C\" __global__ void __launch_bounds__(" << bs << ", 2) hy_taylor(const hy_kargs a)\n{\n";
    src << "__shared__ double lds_slab[" << static_cast<std::uint64_t>(spw) * slab_stride << "];\n";
    src << "__shared__ double lds_jet[" << jet_doubles << "];\n";
    src << "__shared__ double lds_xm[" << 2u * spw * 4u << "];\n__shared__ u64 lds_base;\n";
    src << "const unsigned lane = threadIdx.x & 63u;\n";
    src << "const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);\n";
    src << "const unsigned l = lane % " << L << "u;\nconst unsigned q = lane / " << L << "u;\n";
    src << "const u64 N = a.N;\n";
    src << "double *const slab = lds_slab + q * " << slab_stride << "u;\n";
    src << "double *const jetw = lds_jet + q * " << n_colp << "u;\n";
    src << "double *const xm_mine = lds_xm + (role * " << spw << "u + q) * 4u;\n";
    src << "const double *const xm_other = lds_xm + ((1u - role) * " << spw << "u + q) * 4u;\n";
    for (std::size_t t = 0; t < utbl.size(); ++t) {
        src << "const unsigned ut" << t << " = hy_utbl[" << t * L << "u + l];\n";
    }
    for (std::size_t t = 0; t < dtbl.size(); ++t) {
        src << "const double dt" << t << " = hy_dtbl[" << t * L << "u + l];\n";
    }
    // Owner pointers of both roles (a lane uses those of its role only).
    for (std::uint32_t r = 0; r < 2u; ++r) {
        const char *rn = (r == 0u) ? "A" : "B";
        for (std::size_t oi = 0; oi < rg[r].owners.size(); ++oi) {
            const auto &ow = rg[r].owners[oi];
            src << "const bool ov" << rn << oi << " = l < " << role_count[r] << "u;\n";
            src << "double *const jw" << rn << oi << " = jetw + (ov" << rn << oi << " ? " << utname(ow.col_tbl) << " : " << n_col
                << "u);\n";
            src << "const double *const jr" << rn << oi << " = jetw + " << utname(ow.col_tbl) << ";\n";
        }
    }
    // Lane slots of the final evaluation: slot h -> role h % 2, column (h * L + l).
    for (std::uint32_t h = 0; h < n_hslots; ++h) {
        src << "double *const hc" << h << " = jetw + ((" << h * L << "u + l < " << n_col << "u) ? " << h * L << "u + l : "
            << n_col << "u);\n";
    }
    src << R"HIP(
for (;;) {
// The workgroup pulls the next group of systems from the device-side work queue.
__syncthreads();
if (threadIdx.x == 0u) lds_base = atomicAdd((u64 *)(a.counters + 2), (u64)SPW);
__syncthreads();
u64 base = lds_base;
base = ((u64)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) | (u64)__builtin_amdgcn_readfirstlane((unsigned)base);
if (base >= N) break;
// NOTE: lanes beyond the end of the ensemble replicate the last system (no side effects).
const bool live = (base + q) < N;
const u64 s = live ? (base + q) : (N - 1u);
double t_hi = a.time_hi[s], t_lo = a.time_lo[s];
)HIP";
    for (std::uint32_t r = 0; r < 2u; ++r) {
        const char *rn = (r == 0u) ? "A" : "B";
        src << (r == 0u ? "if (role == 0u) {\n" : "} else {\n");
        for (std::size_t oi = 0; oi < rg[r].owners.size(); ++oi) {
            src << "jw" << rn << oi << "[0] = a.state[(u64)" << utname(rg[r].owners[oi].var_tbl) << " * N + s];\n";
        }
    }
    src << "}\n__syncthreads();\n";
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
// Uniform step loop: the two wavefronts compute the per-system bookkeeping redundantly (identical inputs -> identical
// decisions) and leave together when the 4 systems of the workgroup are done (a finished system keeps taking
// zero-length steps with its bookkeeping frozen).
bool fin = false;
int nf_seen = 0;
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
lim = fin ? 0.0 : lim;
if (role == 0u) {
)HIP";
    src << role_body[0] << "} else {\n" << role_body[1] << "}\n";
    src << R"HIP(
__syncthreads();
const double m0 = hy_max(xm_mine[0], xm_other[0]), mo = hy_max(xm_mine[1], xm_other[1]), mom1 = hy_max(xm_mine[2], xm_other[2]);
)HIP";
    src << "const double num_rho = (m0 <= 1.0) ? 1.0 : m0;\n";
    src << "const double rho_o = hy_root(num_rho / mo, " << fp_literal(1. / static_cast<double>(order)) << ");\n";
    src << "const double rho_om1 = hy_root(num_rho / mom1, " << fp_literal(1. / static_cast<double>(order - 1u)) << ");\n";
    src << "const double rho_m = hy_min(rho_o, rho_om1);\n";
    src << "double h = rho_m * " << fp_literal(rhofac(order)) << ";\n";
    src << "h = hy_min(h, fabs(lim));\nh = (lim < 0.0) ? -h : h;\n";
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
    // Final evaluation: slot h belongs to role h % 2.
    const auto emit_horner = [&](std::uint32_t c) {
        src << "{\nconst double *c = hc" << c << ";\n";
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
        src << "xn" << c << " = res;\nnfi |= !hy_finite(res) ? 1 : 0;\n}\n";
    };
    for (std::uint32_t c = 0; c < n_hslots; ++c) {
        src << "double xn" << c << " = 0.0;\n";
    }
    for (std::uint32_t r = 0; r < 2u; ++r) {
        src << (r == 0u ? "if (role == 0u) {\n" : "} else {\n");
        for (std::uint32_t c = r; c < n_hslots; c += 2u) {
            emit_horner(c);
        }
    }
    src << "}\n";
    for (std::uint32_t m = 1; m < L; m *= 2u) {
        src << "nfi |= __shfl_xor(nfi, " << m << ", 64);\n";
    }
    src << "xm_mine[3] = (double)nfi;\n";
    // Taylor coefficients on request (every lane stores: replicated lanes write identical values).
    src << "if (a.tc != nullptr) {\n";
    for (std::uint32_t r = 0; r < 2u; ++r) {
        const char *rn = (r == 0u) ? "A" : "B";
        src << (r == 0u ? "if (role == 0u) {\n" : "} else {\n");
        for (std::size_t oi = 0; oi < rg[r].owners.size(); ++oi) {
            src << "{\nconst double *c = jr" << rn << oi << ";\ndouble *tcp = a.tc + ((u64)" << utname(rg[r].owners[oi].var_tbl)
                << " * " << (order + 1u) << "u) * N + s;\n#pragma nounroll\nfor (unsigned k = 0; k <= " << order
                << "u; ++k) {\n*tcp = c[(u64)k * " << kstride << "u];\ntcp += N;\n}\n}\n";
        }
    }
    src << "}\n}\n";
    src << "__syncthreads();\n";
    src << "nfi |= (xm_other[3] != 0.0) ? 1 : 0;\n";
    for (std::uint32_t r = 0; r < 2u; ++r) {
        src << (r == 0u ? "if (role == 0u) {\n" : "} else {\n");
        for (std::uint32_t c = r; c < n_hslots; c += 2u) {
            src << "hc" << c << "[0] = fin ? hc" << c << "[0] : xn" << c << ";\n";
        }
    }
    src << "}\n__syncthreads();\n";
    src << R"HIP(
{
    const bool nf = nfi != 0;
    const i64 oc_new = nf ? HY_OC_ERR_NF_STATE : ((h == lim) ? HY_OC_TIME_LIMIT : HY_OC_SUCCESS);
    bool done = nf | (a.mode != 1);
    const u64 ns_new = n_steps + ((!done & (h != 0.0)) ? 1u : 0u);
    const bool upd = !done & (oc_new == HY_OC_SUCCESS);
    const double ah = fabs(h);
    const double mn_new = upd ? hy_min(min_h, ah) : min_h;
    const double mx_new = upd ? hy_max(max_h, ah) : max_h;
    done |= (h == rem.hi);
    hy_df tnew; tnew.hi = nt_hi; tnew.lo = nt_lo;
    const hy_df rem_new = hy_df_sub(tfin, tnew);
    const u64 it_new = iter + 1u;
    const bool sl = !done & (it_new == a.max_steps);
    const i64 oc_fin = sl ? HY_OC_STEP_LIMIT : oc_new;
    done |= sl;
    nf_seen |= (!fin & nf) ? 1 : 0;
    t_hi = fin ? t_hi : nt_hi;
    t_lo = fin ? t_lo : nt_lo;
    last_h = fin ? last_h : h;
    outcome = fin ? outcome : oc_fin;
    n_steps = fin ? n_steps : ns_new;
    min_h = fin ? min_h : mn_new;
    max_h = fin ? max_h : mx_new;
    rem.hi = fin ? rem.hi : rem_new.hi;
    rem.lo = fin ? rem.lo : rem_new.lo;
    iter = fin ? iter : it_new;
    fin = fin | done;
}
if (__builtin_amdgcn_ballot_w64(!fin) == 0ull) break;
}
if (nf_seen != 0 && l == 0u && live && role == 0u) atomicAdd(a.counters, 1u);
)HIP";
    for (std::uint32_t r = 0; r < 2u; ++r) {
        const char *rn = (r == 0u) ? "A" : "B";
        src << (r == 0u ? "if (role == 0u) {\n" : "} else {\n");
        for (std::size_t oi = 0; oi < rg[r].owners.size(); ++oi) {
            src << "if (ov" << rn << oi << " && live) a.state[(u64)" << utname(rg[r].owners[oi].var_tbl) << " * N + s] = jw" << rn
                << oi << "[0];\n";
        }
    }
    src << "}\n";
    src << R"HIP(
if (l == 0u && live && role == 0u) {
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
    // NOTE: the launch geometry counts 2 wavefronts x 16 lanes per system.
    ret.lanes_per_system = 2u * L;
    ret.lds_bytes = 0;
    ret.mode = emit_mode::cluster;
    ret.scratch_per_wave = 0;
    ret.persistent = true;
    ret.tc_optional = true;
    ret.notes = "cluster mode v4 (wave roles, 2 wavefronts per SIMD): " + std::to_string(nc) + " clusters of "
                + std::to_string(pl.clusters[0].size()) + " nodes split over 2 wavefronts, L=" + std::to_string(L) + ", "
                + std::to_string(ns) + " LDS slots x2, " + std::to_string(2u * order + 2u) + " barriers per step, "
                + std::to_string(utbl.size()) + " slot tables, jets in LDS";
    return ret;
}

} // namespace heyoka_amd
