// Model right-hand sides. The construction sequence (which products are formed, and in which
// order the operands appear) follows the reference so that constant folding and the resulting
// decomposition match (reference: src/model/nbody.cpp:53-174, src/model/pendulum.cpp:23-33).
#include "model.hpp"

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

expression r2_of(const expression &dx, const expression &dy, const expression &dz)
{
    return sum({pow(dx, expression{2.}), pow(dy, expression{2.}), pow(dz, expression{2.})});
}

} // namespace

std::vector<std::pair<expression, expression>> nbody_impl(std::uint32_t n, const expression &Gconst,
                                                          const std::vector<expression> &masses)
{
    nbody_checks(n, masses);

    std::vector<expression> x, y, z, vx, vy, vz;
    for (std::uint32_t i = 0; i < n; ++i) {
        const auto s = std::to_string(i);
        x.emplace_back("x_" + s);
        y.emplace_back("y_" + s);
        z.emplace_back("z_" + s);
        vx.emplace_back("vx_" + s);
        vy.emplace_back("vy_" + s);
        vz.emplace_back("vz_" + s);
    }

    std::vector<std::pair<expression, expression>> retval;
    std::vector<std::vector<expression>> x_acc(n), y_acc(n), z_acc(n);

    const auto n_massive = static_cast<std::uint32_t>(masses.size());

    for (std::uint32_t i = 0; i < n_massive; ++i) {
        retval.emplace_back(x[i], vx[i]);
        retval.emplace_back(y[i], vy[i]);
        retval.emplace_back(z[i], vz[i]);

        for (std::uint32_t j = i + 1u; j < n; ++j) {
            const auto diff_x = x[j] - x[i];
            const auto diff_y = y[j] - y[i];
            const auto diff_z = z[j] - z[i];

            const auto r_m3 = pow(r2_of(diff_x, diff_y, diff_z), expression{-3. / 2});

            const auto j_massive = j < n_massive;
            // Grouping that maximises constant folding, when masses and G are numbers.
            const auto opt_grouping
                = j_massive && masses[j].is_number() && masses[j].num() != 0 && Gconst.is_number();

            if (opt_grouping) {
                const auto fac_j = Gconst * masses[j] * r_m3;
                const auto c_ij = -masses[i] / masses[j];

                // j on i.
                x_acc[i].push_back(diff_x * fac_j);
                y_acc[i].push_back(diff_y * fac_j);
                z_acc[i].push_back(diff_z * fac_j);

                // i on j.
                x_acc[j].push_back(x_acc[i].back() * c_ij);
                y_acc[j].push_back(y_acc[i].back() * c_ij);
                z_acc[j].push_back(z_acc[i].back() * c_ij);
            } else {
                const auto G_r_m3 = Gconst * r_m3;

                const auto fac_i = -masses[i] * G_r_m3;
                x_acc[j].push_back(diff_x * fac_i);
                y_acc[j].push_back(diff_y * fac_i);
                z_acc[j].push_back(diff_z * fac_i);

                if (j_massive) {
                    const auto fac_j = masses[j] * G_r_m3;
                    x_acc[i].push_back(diff_x * fac_j);
                    y_acc[i].push_back(diff_y * fac_j);
                    z_acc[i].push_back(diff_z * fac_j);
                }
            }
        }

        retval.emplace_back(vx[i], sum(x_acc[i]));
        retval.emplace_back(vy[i], sum(y_acc[i]));
        retval.emplace_back(vz[i], sum(z_acc[i]));
    }

    for (auto i = n_massive; i < n; ++i) {
        retval.emplace_back(x[i], vx[i]);
        retval.emplace_back(y[i], vy[i]);
        retval.emplace_back(z[i], vz[i]);

        retval.emplace_back(vx[i], sum(x_acc[i]));
        retval.emplace_back(vy[i], sum(y_acc[i]));
        retval.emplace_back(vz[i], sum(z_acc[i]));
    }

    return retval;
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

} // namespace heyoka_amd::model::detail
