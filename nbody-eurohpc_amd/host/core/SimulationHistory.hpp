// Per-iteration metrics of a tracked run: the host-side counterpart of the reference's
// SimulationHistory<T> (reference src/common/core/SimulationHistory.hpp:10-49) with the same accessor
// names and the same CSV layout (SimulationHistory.cpp:103-121):
//     iteration,energy,ang_momentum,density_center_x,density_center_y,density_center_z
// printed with max_digits10 significant digits.  The reference needs a device twin
// (GPUSimulationHistory) because its energy reduction ends in device memory; here the O(N) sums are
// done in fp64 on the host (murbhip_energy / murbhip_moments), so one class is enough.
#ifndef SIMULATION_HISTORY_HPP_
#define SIMULATION_HISTORY_HPP_

#include <array>
#include <cstddef>
#include <fstream>
#include <iomanip>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

template <typename T> class SimulationHistory {
  public:
    struct Row {
        T energy{};
        T angMomentum{};
        std::array<T, 3> densityCenter{};
    };

    SimulationHistory() = default;
    explicit SimulationHistory(int numIterations) { setNumIterations(numIterations); }
    virtual ~SimulationHistory() = default;

    void setNumIterations(int numIterations) { rows.resize(numIterations < 0 ? 0 : (std::size_t)numIterations); }
    int getNumIterations() const { return (int)rows.size(); }

    T getEnergyAt(int it) const { return rows.at(it).energy; }
    void setEnergyAt(int it, T v) { rows.at(it).energy = v; }
    T getAngMomentumAt(int it) const { return rows.at(it).angMomentum; }
    void setAngMomentumAt(int it, T v) { rows.at(it).angMomentum = v; }
    const std::array<T, 3> &getDensityCenterAt(int it) const { return rows.at(it).densityCenter; }
    void setDensityCenterAt(int it, const std::array<T, 3> &v) { rows.at(it).densityCenter = v; }

    std::vector<T> getAllEnergy() const { return column([](const Row &r) { return r.energy; }); }
    std::vector<T> getAllAngMomentum() const { return column([](const Row &r) { return r.angMomentum; }); }
    std::vector<std::array<T, 3>> getAllDensityCenter() const
    {
        std::vector<std::array<T, 3>> out;
        for (const Row &r : rows) out.push_back(r.densityCenter);
        return out;
    }
    void setAllEnergy(const std::vector<T> &v) { assign(v, [](Row &r, T x) { r.energy = x; }); }
    void setAllAngMomentum(const std::vector<T> &v) { assign(v, [](Row &r, T x) { r.angMomentum = x; }); }
    void setAllDensityCenter(const std::vector<std::array<T, 3>> &v)
    {
        rows.resize(v.size());
        for (std::size_t i = 0; i < v.size(); ++i) rows[i].densityCenter = v[i];
    }

    // throws std::runtime_error when the file cannot be opened, like the reference (SimulationHistory.cpp:106-108)
    void saveMetricsToCSV(const std::string &filePath) const
    {
        std::ofstream out(filePath);
        if (!out.is_open())
            throw std::runtime_error("SimulationHistory::saveMetricsToCSV: cannot open file '" + filePath + "'");
        out << "iteration,energy,ang_momentum,density_center_x,density_center_y,density_center_z\n"
            << std::setprecision(std::numeric_limits<T>::max_digits10);
        for (std::size_t i = 0; i < rows.size(); ++i) {
            const Row &r = rows[i];
            out << i << ',' << r.energy << ',' << r.angMomentum << ',' << r.densityCenter[0] << ',' << r.densityCenter[1]
                << ',' << r.densityCenter[2] << '\n';
        }
    }

  protected:
    std::vector<Row> rows;

  private:
    template <typename F> std::vector<T> column(F get) const
    {
        std::vector<T> out;
        out.reserve(rows.size());
        for (const Row &r : rows) out.push_back(get(r));
        return out;
    }
    template <typename F> void assign(const std::vector<T> &v, F set)
    {
        rows.resize(v.size());
        for (std::size_t i = 0; i < v.size(); ++i) set(rows[i], v[i]);
    }
};

#endif
