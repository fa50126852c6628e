// Registry of node rules (see node_rule.hpp).
#include "node_rule.hpp"

#include <cctype>
#include <deque>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace heyoka_amd
{

namespace
{

struct registry {
    std::mutex mtx;
    // NOTE: a deque - references to registered rules stay valid while others are added.
    std::deque<node_rule> rules;
};

registry &reg()
{
    static registry r;
    return r;
}

bool valid_identifier(const std::string &s)
{
    if (s.empty() || (std::isalpha(static_cast<unsigned char>(s[0])) == 0 && s[0] != '_')) {
        return false;
    }
    for (const auto c : s) {
        if (std::isalnum(static_cast<unsigned char>(c)) == 0 && c != '_') {
            return false;
        }
    }
    return true;
}

} // namespace

std::uint32_t register_node_rule(node_rule r)
{
    // The built-in rules (kepF, kepDE, pi) go in first, whoever registers first: a user rule of one of those names must be
    // refused HERE - registered ahead of them it would make the built-in registration throw on every later use.
    ensure_builtin_rules();
    if (!valid_identifier(r.name)) {
        throw std::invalid_argument("Invalid name for a node rule: '" + r.name + "' is not an identifier");
    }
    for (int k = 0; k <= static_cast<int>(func_kind::custom); ++k) {
        if (r.name == func_kind_name(static_cast<func_kind>(k))) {
            throw std::invalid_argument("Cannot register the node rule '" + r.name + "': the name of a built-in function");
        }
    }
    // (No arguments: a named constant - reference: the constant class, include/heyoka/math/constants.hpp:75-117.)
    if (r.n_args > 8u) {
        throw std::invalid_argument("A node rule can have at most 8 arguments, but the rule '" + r.name + "' has "
                                    + std::to_string(r.n_args));
    }
    for (const auto *fn : {"_order0", "_orderk"}) {
        if (r.hip_source.find("hy_rule_" + r.name + fn) == std::string::npos) {
            throw std::invalid_argument("The HIP source of the node rule '" + r.name + "' does not define hy_rule_" + r.name
                                        + fn + "()");
        }
    }
    if (r.deps.size() > 8u) {
        throw std::invalid_argument("A node rule can have at most 8 hidden dependencies");
    }
    auto &g = reg();
    std::lock_guard<std::mutex> lock(g.mtx);
    for (const auto &o : g.rules) {
        if (o.name == r.name) {
            throw std::invalid_argument("A node rule named '" + r.name + "' has been registered already");
        }
    }
    g.rules.push_back(std::move(r));
    return static_cast<std::uint32_t>(g.rules.size());
}

std::uint32_t node_rule_id(const std::string &name)
{
    ensure_builtin_rules();
    auto &g = reg();
    std::lock_guard<std::mutex> lock(g.mtx);
    for (std::size_t i = 0; i < g.rules.size(); ++i) {
        if (g.rules[i].name == name) {
            return static_cast<std::uint32_t>(i + 1u);
        }
    }
    return 0;
}

const node_rule *find_node_rule(const std::string &name)
{
    const auto id = node_rule_id(name);
    return id == 0u ? nullptr : &get_node_rule(id);
}

const node_rule &get_node_rule(std::uint32_t id)
{
    auto &g = reg();
    std::lock_guard<std::mutex> lock(g.mtx);
    if (id == 0u || id > g.rules.size()) {
        throw std::invalid_argument("Invalid node rule id: " + std::to_string(id));
    }
    return g.rules[id - 1u];
}

expression custom_func(const std::string &name, std::vector<expression> args)
{
    const auto id = node_rule_id(name);
    if (id == 0u) {
        // Reference: a function without an implementation raises at the point of use (func.hpp:266-267).
        throw not_implemented_error("Taylor diff is not implemented for the function '" + name + "'");
    }
    const auto &r = get_node_rule(id);
    if (args.size() != r.n_args) {
        throw std::invalid_argument("The function '" + name + "' takes " + std::to_string(r.n_args)
                                    + " argument(s), but " + std::to_string(args.size()) + " were provided");
    }
    if (r.fold) {
        expression out;
        if (r.fold(args, out)) {
            return out;
        }
    }
    return expression{func(func_kind::custom, std::move(args), id)};
}

std::string node_rules_device_source(const std::vector<std::uint32_t> &rule_ids)
{
    if (rule_ids.empty()) {
        return {};
    }
    std::ostringstream os;
    os << R"HIP(
// ---- node rules (heyoka_amd/csrc/node_rule.hpp): a jet is a strided view of Taylor coefficients ----
struct hy_jet {
    const double *p; // coefficient j at p[j * s]
    unsigned s;      // stride, in doubles
    unsigned n;      // number of valid coefficients (orders 0 .. n - 1); beyond: 0
};
__device__ __forceinline__ double hy_jc(const hy_jet &x, unsigned j)
{
    return j < x.n ? x.p[(u64)j * x.s] : 0.0;
}
)HIP";
    std::vector<std::uint32_t> done;
    for (const auto id : rule_ids) {
        bool dup = false;
        for (const auto d : done) {
            dup = dup || d == id;
        }
        if (!dup) {
            done.push_back(id);
            const auto &r = get_node_rule(id);
            os << "// rule: " << r.name << "\n" << r.hip_source << "\n";
            // What the generators call for order 0: the rule's function behind an out-of-line frame with the arguments by
            // value. Order-0 rules are where library calls and data-dependent loops live (a Newton iteration, the
            // large-argument path of sin()): inlined into a stepper with hundreds of live registers they would put
            // divergent regions into it ("Toolchain notes": live-range splits inside exec-masked blocks).
            os << "static __device__ __attribute__((noinline)) double hy_rule_" << r.name << "_value(";
            for (std::uint32_t i = 0; i < r.n_args; ++i) {
                os << (i == 0u ? "" : ", ") << "double x" << i;
            }
            if (r.n_args == 0u) {
                os << ")\n{\n    return hy_rule_" << r.name << "_order0(nullptr);\n}\n";
                continue;
            }
            os << ")\n{\n    const double x[] = {";
            for (std::uint32_t i = 0; i < r.n_args; ++i) {
                os << (i == 0u ? "" : ", ") << "x" << i;
            }
            os << "};\n    return hy_rule_" << r.name << "_order0(x);\n}\n";
        }
    }
    return os.str();
}

} // namespace heyoka_amd
