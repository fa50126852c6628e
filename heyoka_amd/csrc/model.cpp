// Model right-hand sides. The construction sequence (which products are formed, and in which
// order the operands appear) follows the reference so that constant folding and the resulting
// decomposition match (reference: src/model/nbody.cpp:53-174, src/model/pendulum.cpp:23-33).
#include "model.hpp"

#include <cmath>
#include <cstddef>
#include <sstream>
#include <stdexcept>
#include <string>

namespace heyoka_amd::model::detail
{

namespace
{

void nbody_checks(std::uint32_t n, const std::vector<expression> &masses)
{
    if (n < 2u) {
        throw std::invalid_argument("Cannot construct an N-body system with N == " + std::to_string(n)
                                    + ": at least 2 bodies are needed");
    }
    if (masses.size() > n) {
        throw std::invalid_argument("In an N-body system the number of particles with mass ("
                                    + std::to_string(masses.size())
                                    + ") cannot be greater than the total number of particles (" + std::to_string(n)
                                    + ")");
    }
}

// "{}"-style formatting of a double for error messages (-1 -> "-1").
std::string detail_fmt_double(double v)
{
    std::ostringstream oss;
    oss.precision(17);
    oss << v;
    // Shortest representation which round-trips.
    for (int prec = 1; prec <= 17; ++prec) {
        std::ostringstream t;
        t.precision(prec);
        t << v;
        if (std::stod(t.str()) == v) {
            return t.str();
        }
    }
    return oss.str();
}

expression r2_of(const expression &dx, const expression &dy, const expression &dz)
{
    return sum({pow(dx, expression{2.}), pow(dy, expression{2.}), pow(dz, expression{2.})});
}

} // namespace

// Newtonian N-body right-hand side (the model of src/model/nbody.cpp:53-174; what must agree with it is the RESULT -
// the expression of every acceleration term and the order of the terms inside each sum, since the decomposition, its CSE
// and the sizes pinned by test/model_nbody.cpp follow from them - not the text).
//
// Built axis by axis: bodies carry pos[3] / vel[3], every unordered pair {i < j} with a massive first body contributes one
// term per axis to the acceleration lists of its two bodies, in the order of the pairs. Two forms of the pair term:
//  * "scaled" (both bodies massive, m_j and G numerical): the term on i is d * (G m_j r^-3) and the term on j is that very
//    expression times -(m_i / m_j), so that the constants fold into two numbers per pair;
//  * general: the terms are d * (-m_i (G r^-3)) on j and, if j is massive, d * (m_j (G r^-3)) on i.
std::vector<std::pair<expression, expression>> nbody_impl(std::uint32_t n, const expression &Gconst,
                                                          const std::vector<expression> &masses)
{
    nbody_checks(n, masses);

    static constexpr const char *axis_name[3] = {"x_", "y_", "z_"};
    struct body {
        expression pos[3], vel[3];
        std::vector<expression> acc[3]; // terms of the acceleration, per axis
    };
    std::vector<body> bodies(n);
    for (std::uint32_t b = 0; b < n; ++b) {
        for (int c = 0; c < 3; ++c) {
            const auto name = axis_name[c] + std::to_string(b);
            bodies[b].pos[c] = expression{name};
            bodies[b].vel[c] = expression{"v" + name};
        }
    }

    const auto n_massive = static_cast<std::uint32_t>(masses.size());
    const auto is_massive = [&](std::uint32_t b) { return b < n_massive; };
    // A pair takes the scaled form when the mass of its second body is a non-zero number and G is a number too.
    const auto scaled_form = [&](std::uint32_t j) {
        return is_massive(j) && masses[j].is_number() && masses[j].num() != 0 && Gconst.is_number();
    };

    for (std::uint32_t i = 0; i < n_massive; ++i) {
        for (std::uint32_t j = i + 1u; j < n; ++j) {
            auto &bi = bodies[i];
            auto &bj = bodies[j];
            expression d[3];
            for (int c = 0; c < 3; ++c) {
                d[c] = bj.pos[c] - bi.pos[c];
            }
            const auto inv_r3 = pow(r2_of(d[0], d[1], d[2]), expression{-3. / 2});

            if (scaled_form(j)) {
                const auto on_i = Gconst * masses[j] * inv_r3;
                const auto reaction = -masses[i] / masses[j];
                for (int c = 0; c < 3; ++c) {
                    const auto term = d[c] * on_i;
                    bi.acc[c].push_back(term);
                    bj.acc[c].push_back(term * reaction);
                }
            } else {
                const auto g_inv_r3 = Gconst * inv_r3;
                const auto on_j = -masses[i] * g_inv_r3;
                for (int c = 0; c < 3; ++c) {
                    bj.acc[c].push_back(d[c] * on_j);
                }
                if (is_massive(j)) {
                    const auto on_i = masses[j] * g_inv_r3;
                    for (int c = 0; c < 3; ++c) {
                        bi.acc[c].push_back(d[c] * on_i);
                    }
                }
            }
        }
    }

    // x' = v for the three axes of a body, then v' = sum of its acceleration terms; bodies in order (the massive ones
    // come first by construction of the masses argument).
    std::vector<std::pair<expression, expression>> sys;
    sys.reserve(static_cast<std::size_t>(n) * 6u);
    for (auto &b : bodies) {
        for (int c = 0; c < 3; ++c) {
            sys.emplace_back(b.pos[c], b.vel[c]);
        }
        for (int c = 0; c < 3; ++c) {
            sys.emplace_back(b.vel[c], sum(b.acc[c]));
        }
    }
    return sys;
}

expression nbody_potential_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses)
{
    nbody_checks(n, masses);
    const auto n_massive = static_cast<std::uint32_t>(masses.size());

    std::vector<expression> x, y, z;
    for (std::uint32_t i = 0; i < n_massive; ++i) {
        const auto s = std::to_string(i);
        x.emplace_back("x_" + s);
        y.emplace_back("y_" + s);
        z.emplace_back("z_" + s);
    }

    std::vector<expression> pot;
    for (std::uint32_t i = 0; i < n_massive; ++i) {
        for (std::uint32_t j = i + 1u; j < n_massive; ++j) {
            pot.push_back(masses[i] * masses[j] / sqrt(r2_of(x[j] - x[i], y[j] - y[i], z[j] - z[i])));
        }
    }

    return -Gconst * sum(pot);
}

expression nbody_energy_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses)
{
    nbody_checks(n, masses);
    const auto n_massive = static_cast<std::uint32_t>(masses.size());

    std::vector<expression> kin;
    for (std::uint32_t i = 0; i < n_massive; ++i) {
        const auto s = std::to_string(i);
        kin.push_back(masses[i]
                      * r2_of(expression{"vx_" + s}, expression{"vy_" + s}, expression{"vz_" + s}));
    }

    return expression{.5} * sum(kin) + nbody_potential_impl(n, Gconst, masses);
}

std::vector<std::pair<expression, expression>> pendulum_impl(const expression &gconst, const expression &length)
{
    const expression x{"x"}, v{"v"};
    return {{x, v}, {v, -gconst / length * sin(x)}};
}

expression pendulum_energy_impl(const expression &gconst, const expression &length)
{
    const expression x{"x"}, v{"v"};
    return expression{.5} * pow(length, expression{2.}) * pow(v, expression{2.})
           + gconst * length * (expression{1.} - cos(x));
}

// ---- N+1 bodies in the frame of body 0 (reference: src/model/nbody.cpp:236-325). ----
// Bodies 1..n-1 with coordinates relative to body 0: the direct attraction of body 0 on the (reduced) pair, the
// mutual attractions of the massive companions and the indirect terms from the acceleration of the origin.
std::vector<std::pair<expression, expression>> np1body_impl(std::uint32_t n, const expression &Gconst,
                                                            const std::vector<expression> &masses)
{
    nbody_checks(n, masses);

    const auto nm1 = n - 1u;
    std::vector<expression> x, y, z, vx, vy, vz;
    for (std::uint32_t i = 1; i <= nm1; ++i) {
        const auto s = std::to_string(i);
        x.emplace_back("x_" + s);
        y.emplace_back("y_" + s);
        z.emplace_back("z_" + s);
        vx.emplace_back("vx_" + s);
        vy.emplace_back("vy_" + s);
        vz.emplace_back("vz_" + s);
    }

    // r_i / |r_i|^3 for every body.
    std::vector<expression> xr3, yr3, zr3;
    for (std::uint32_t i = 0; i < nm1; ++i) {
        const auto rm3 = pow(r2_of(x[i], y[i], z[i]), expression{-3. / 2});
        xr3.push_back(x[i] * rm3);
        yr3.push_back(y[i] * rm3);
        zr3.push_back(z[i] * rm3);
    }

    const auto n_massive = static_cast<std::uint32_t>(masses.size());
    const auto mass_or_zero = [&](std::uint32_t idx) { return idx < n_massive ? masses[idx] : expression{0.}; };

    std::vector<std::pair<expression, expression>> retval;
    for (std::uint32_t i = 0; i < nm1; ++i) {
        retval.emplace_back(x[i], vx[i]);
        retval.emplace_back(y[i], vy[i]);
        retval.emplace_back(z[i], vz[i]);

        std::vector<expression> ax, ay, az;

        // Two-body term with the combined mass.
        const auto mu_0i = -Gconst * (mass_or_zero(0) + mass_or_zero(i + 1u));
        ax.push_back(mu_0i * xr3[i]);
        ay.push_back(mu_0i * yr3[i]);
        az.push_back(mu_0i * zr3[i]);

        // Massive companions: direct + indirect term each.
        for (std::uint32_t j = 0; n_massive > 0u && j + 1u < n_massive; ++j) {
            if (j == i) {
                continue;
            }
            // NOTE: the difference is always (higher index) - (lower index), so that the pair (i, j) and the pair
            // (j, i) share the subexpressions.
            const bool fwd = j > i;
            const auto dx = fwd ? x[j] - x[i] : x[i] - x[j];
            const auto dy = fwd ? y[j] - y[i] : y[i] - y[j];
            const auto dz = fwd ? z[j] - z[i] : z[i] - z[j];
            const auto drm3 = pow(r2_of(dx, dy, dz), expression{-1.5});
            const auto mu_j = Gconst * masses[j + 1u];

            const auto aij_x = mu_j * (dx * drm3);
            const auto aij_y = mu_j * (dy * drm3);
            const auto aij_z = mu_j * (dz * drm3);
            ax.push_back(fwd ? aij_x : -aij_x);
            ay.push_back(fwd ? aij_y : -aij_y);
            az.push_back(fwd ? aij_z : -aij_z);

            ax.push_back(-mu_j * xr3[j]);
            ay.push_back(-mu_j * yr3[j]);
            az.push_back(-mu_j * zr3[j]);
        }

        retval.emplace_back(vx[i], sum(ax));
        retval.emplace_back(vy[i], sum(ay));
        retval.emplace_back(vz[i], sum(az));
    }

    return retval;
}

// Reference: src/model/nbody.cpp:327-372.
expression np1body_potential_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses)
{
    nbody_checks(n, masses);
    if (masses.empty()) {
        return expression{0.};
    }

    const auto nc = static_cast<std::uint32_t>(masses.size()) - 1u;
    std::vector<expression> x, y, z;
    for (std::uint32_t i = 1; i <= nc; ++i) {
        const auto s = std::to_string(i);
        x.emplace_back("x_" + s);
        y.emplace_back("y_" + s);
        z.emplace_back("z_" + s);
    }

    std::vector<expression> terms;
    for (std::uint32_t i = 0; i < nc; ++i) {
        terms.push_back(masses[0] * masses[i + 1u] / sqrt(r2_of(x[i], y[i], z[i])));
    }
    for (std::uint32_t i = 0; i < nc; ++i) {
        for (std::uint32_t j = i + 1u; j < nc; ++j) {
            terms.push_back(masses[i + 1u] * masses[j + 1u] / sqrt(r2_of(x[j] - x[i], y[j] - y[i], z[j] - z[i])));
        }
    }

    return -Gconst * sum(terms);
}

// Reference: src/model/nbody.cpp:374-445. The kinetic energy is written in the barycentric frame: the velocity of
// body 0 follows from the conservation of the linear momentum.
expression np1body_energy_impl(std::uint32_t n, const expression &Gconst, const std::vector<expression> &masses)
{
    nbody_checks(n, masses);
    if (masses.empty()) {
        return expression{0.};
    }

    const auto nc = static_cast<std::uint32_t>(masses.size()) - 1u;
    std::vector<expression> vx, vy, vz;
    for (std::uint32_t i = 1; i <= nc; ++i) {
        const auto s = std::to_string(i);
        vx.emplace_back("vx_" + s);
        vy.emplace_back("vy_" + s);
        vz.emplace_back("vz_" + s);
    }

    std::vector<expression> px, py, pz, mtot{masses[0]};
    for (std::uint32_t i = 0; i < nc; ++i) {
        px.push_back(masses[i + 1u] * vx[i]);
        py.push_back(masses[i + 1u] * vy[i]);
        pz.push_back(masses[i + 1u] * vz[i]);
        mtot.push_back(masses[i + 1u]);
    }
    const auto M = sum(mtot);
    const auto u0x = sum(px) / -M;
    const auto u0y = sum(py) / -M;
    const auto u0z = sum(pz) / -M;

    std::vector<expression> kin{masses[0] * r2_of(u0x, u0y, u0z)};
    for (std::uint32_t i = 0; i < nc; ++i) {
        kin.push_back(masses[i + 1u] * r2_of(vx[i] + u0x, vy[i] + u0y, vz[i] + u0z));
    }

    return expression{.5} * sum(kin) + np1body_potential_impl(n, Gconst, masses);
}

// ---- Circular restricted three-body problem in the synodic frame, canonical momenta (src/model/cr3bp.cpp). ----
namespace
{

void cr3bp_check_mu(const expression &mu)
{
    if (mu.is_number()) {
        const auto v = mu.num();
        if (!std::isfinite(v) || v <= 0 || v >= .5) {
            throw std::invalid_argument("The 'mu' parameter in a CR3BP must be in the range (0, "
                                        "0.5), but a value of "
                                        + detail_fmt_double(v) + " was provided instead");
        }
    }
}

// The squared distances from the two primaries and the offsets along x.
struct cr3bp_geom {
    expression dx1, dx2, r1_2, r2_2;
};

cr3bp_geom cr3bp_geometry(const expression &mu)
{
    const expression x{"x"}, y{"y"}, z{"z"};
    cr3bp_geom g;
    g.dx1 = x - mu;
    g.dx2 = g.dx1 + expression{1.};
    const auto yz2 = pow(y, expression{2.}) + pow(z, expression{2.});
    g.r1_2 = pow(g.dx1, expression{2.}) + yz2;
    g.r2_2 = pow(g.dx2, expression{2.}) + yz2;
    return g;
}

} // namespace

std::vector<std::pair<expression, expression>> cr3bp_impl(const expression &mu)
{
    cr3bp_check_mu(mu);

    const expression px{"px"}, py{"py"}, pz{"pz"}, x{"x"}, y{"y"}, z{"z"};
    const auto g = cr3bp_geometry(mu);

    const auto g1 = (expression{1.} - mu) * pow(g.r1_2, expression{-3. / 2});
    const auto g2 = mu * pow(g.r2_2, expression{-3. / 2});
    const auto g12 = g1 + g2;

    return {{x, px + y},
            {y, py - x},
            {z, pz},
            {px, py - g1 * g.dx1 - g2 * g.dx2},
            {py, -px - g12 * y},
            {pz, -g12 * z}};
}

expression cr3bp_jacobi_impl(const expression &mu)
{
    cr3bp_check_mu(mu);

    const expression px{"px"}, py{"py"}, pz{"pz"}, x{"x"}, y{"y"};
    const auto g = cr3bp_geometry(mu);

    const auto g1 = (expression{1.} - mu) / sqrt(g.r1_2);
    const auto g2 = mu / sqrt(g.r2_2);
    const auto kin = expression{.5} * (pow(px, expression{2.}) + pow(py, expression{2.}) + pow(pz, expression{2.}));

    return kin + y * px - x * py - g1 - g2;
}

// ---- Fixed centres of attraction (src/model/fixed_centres.cpp). ----
namespace
{

void fixed_centres_checks(const std::vector<expression> &masses, const std::vector<expression> &positions)
{
    if (positions.size() % 3u != 0u) {
        throw std::invalid_argument(
            "In a fixed centres system the positions vector's size must be a multiple of 3, but instead it is "
            + std::to_string(positions.size()));
    }
    if (positions.size() / 3u != masses.size()) {
        throw std::invalid_argument("In a fixed centres system the number of masses (" + std::to_string(masses.size())
                                    + ") differs from the number of position vectors ("
                                    + std::to_string(positions.size() / 3u) + ")");
    }
}

} // namespace

std::vector<std::pair<expression, expression>>
fixed_centres_impl(const expression &G, const std::vector<expression> &masses, const std::vector<expression> &positions)
{
    fixed_centres_checks(masses, positions);

    const expression x{"x"}, y{"y"}, z{"z"}, vx{"vx"}, vy{"vy"}, vz{"vz"};
    std::vector<expression> ax, ay, az;
    for (std::size_t i = 0; i < masses.size(); ++i) {
        const auto dx = positions[3u * i] - x;
        const auto dy = positions[3u * i + 1u] - y;
        const auto dz = positions[3u * i + 2u] - z;
        const auto m_rm3 = masses[i] * pow(r2_of(dx, dy, dz), expression{-1.5});
        ax.push_back(dx * m_rm3);
        ay.push_back(dy * m_rm3);
        az.push_back(dz * m_rm3);
    }

    return {{x, vx}, {y, vy}, {z, vz}, {vx, G * sum(ax)}, {vy, G * sum(ay)}, {vz, G * sum(az)}};
}

expression fixed_centres_potential_impl(const expression &G, const std::vector<expression> &masses,
                                        const std::vector<expression> &positions)
{
    fixed_centres_checks(masses, positions);

    const expression x{"x"}, y{"y"}, z{"z"};
    std::vector<expression> terms;
    for (std::size_t i = 0; i < masses.size(); ++i) {
        const auto dx = positions[3u * i] - x;
        const auto dy = positions[3u * i + 1u] - y;
        const auto dz = positions[3u * i + 2u] - z;
        terms.push_back(masses[i] / sqrt(r2_of(dx, dy, dz)));
    }

    return -G * sum(terms);
}

expression fixed_centres_energy_impl(const expression &G, const std::vector<expression> &masses,
                                     const std::vector<expression> &positions)
{
    const auto kin = expression{.5} * r2_of(expression{"vx"}, expression{"vy"}, expression{"vz"});
    return kin + fixed_centres_potential_impl(G, masses, positions);
}

// ---- Free particle in a uniformly rotating frame: centrifugal + Coriolis (src/model/rotating.cpp). ----
namespace
{

void rotating_check_omega(const std::vector<expression> &omega)
{
    if (!omega.empty() && omega.size() != 3u) {
        throw std::invalid_argument("In a rotating reference frame model the angular velocity must be a "
                                    "3-dimensional vector, but instead it is a "
                                    + std::to_string(omega.size()) + "-dimensional vector");
    }
}

} // namespace

std::vector<std::pair<expression, expression>> rotating_impl(const std::vector<expression> &omega)
{
    rotating_check_omega(omega);

    const expression x{"x"}, y{"y"}, z{"z"}, vx{"vx"}, vy{"vy"}, vz{"vz"};
    std::vector<expression> ax, ay, az;
    if (!omega.empty()) {
        const auto &p = omega[0], &q = omega[1], &r = omega[2];
        const auto two = expression{2.};

        // -omega x (omega x r), term by term, sharing the products q*x, r*x, q*y, r*z ...
        const auto qx = q * x, rx = r * x, qy = q * y, rz = r * z;
        ax = {q * qx, r * rx, -(p * qy), -(p * rz)};
        ay = {pow(p, two) * y, pow(r, two) * y, -(p * qx), -(q * rz)};
        az = {pow(p, two) * z, pow(q, two) * z, -(p * rx), -(r * qy)};

        // ... and -2 omega x v.
        ax.push_back(expression{-2.} * (q * vz - r * vy));
        ay.push_back(expression{-2.} * (r * vx - p * vz));
        az.push_back(expression{-2.} * (p * vy - q * vx));
    }

    return {{x, vx}, {y, vy}, {z, vz}, {vx, sum(ax)}, {vy, sum(ay)}, {vz, sum(az)}};
}

expression rotating_potential_impl(const std::vector<expression> &omega)
{
    rotating_check_omega(omega);
    if (omega.empty()) {
        return expression{0.};
    }

    const expression x{"x"}, y{"y"}, z{"z"};
    const auto &p = omega[0], &q = omega[1], &r = omega[2];
    // ((omega . r)^2 - omega^2 r^2) / 2
    const auto dot = sum({p * x, q * y, r * z});
    return expression{.5} * (pow(dot, expression{2.}) - r2_of(p, q, r) * r2_of(x, y, z));
}

expression rotating_energy_impl(const std::vector<expression> &omega)
{
    return expression{.5} * r2_of(expression{"vx"}, expression{"vy"}, expression{"vz"})
           + rotating_potential_impl(omega);
}

// ---- Mascon model: fixed centres in a rotating frame (src/model/mascon.cpp). ----
std::vector<std::pair<expression, expression>> mascon_impl(const expression &G, const std::vector<expression> &masses,
                                                           const std::vector<expression> &positions,
                                                           const std::vector<expression> &omega)
{
    auto dyn = fixed_centres_impl(G, masses, positions);
    const auto rot = rotating_impl(omega);
    for (std::size_t i = 3; i < 6u; ++i) {
        dyn[i].second = dyn[i].second + rot[i].second;
    }
    return dyn;
}

expression mascon_energy_impl(const expression &G, const std::vector<expression> &masses,
                              const std::vector<expression> &positions, const std::vector<expression> &omega)
{
    return fixed_centres_energy_impl(G, masses, positions) + rotating_potential_impl(omega);
}

expression mascon_potential_impl(const expression &G, const std::vector<expression> &masses,
                                 const std::vector<expression> &positions, const std::vector<expression> &omega)
{
    return fixed_centres_potential_impl(G, masses, positions) + rotating_potential_impl(omega);
}

} // namespace heyoka_amd::model::detail
