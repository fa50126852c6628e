// Planning structures of the cluster code generators (see hip_emit_cluster.cpp).
#pragma once

#include <cstdint>
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

struct cluster_plan {
    std::uint32_t n_eq = 0, n_u = 0, L = 1, spw = 64;
    std::vector<std::vector<std::uint32_t>> clusters; // member u indices (ascending), all isomorphic
    std::vector<int> cluster_of;                      // per u: cluster id or -1
    std::uint32_t cluster_level = 1;
    // Template-relative descriptions.
    std::vector<std::vector<std::uint32_t>> ext_u;   // [cluster][e] -> u index of the e-th external input
    std::vector<std::pair<std::uint32_t, std::uint32_t>> cst_pos; // (template position, arg index) of per-lane constants
    std::vector<std::vector<double>> cst_val;         // [cluster][slot]
    std::vector<std::uint32_t> out_pos;               // template positions whose value is exported
    std::vector<int> slot_of;                         // per u: LDS slot or -1
    std::uint32_t n_slots = 0, n_dummy = 0;
    std::vector<glue_group> groups;                   // sorted by level
    std::uint32_t max_level = 0;
    std::vector<std::uint32_t> lvl;                   // per u
    // Template positions of the members whose lower-order coefficients are read by a recurrence
    // (they must be kept for the whole step: in registers, or on a tape).
    std::vector<std::uint32_t> stored_pos;
};

// Limits of the target kernel shape: wave mode (one system per group of <= 64 lanes, jets in registers) or
// block mode (one system per workgroup, any number of clusters, jets on a tape).
struct plan_limits {
    std::uint32_t max_clusters = 64;
    bool jets_in_registers = true;
    // Absorb the linear nodes fed by a single cluster into that cluster (fewer glue rounds). Switched off by
    // make_plan() on a second attempt when the absorption makes otherwise isomorphic clusters differ.
    bool absorb_linear = true;
};

// Build the plan; returns an empty string on success, otherwise the reason why cluster mode is not applicable.
std::string make_plan(const taylor_program &p, std::uint32_t order, cluster_plan &pl, const plan_limits &lim = {});

inline bool is_var(const operand &o)
{
    return o.type == operand::kind::uvar;
}

} // namespace heyoka_amd::cluster_detail
