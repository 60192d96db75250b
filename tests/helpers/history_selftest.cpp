// TEST INFRASTRUCTURE — container semantics of host/core/SimulationHistory.hpp, the cases the reference's own
// test pins for its SimulationHistory<float> (src/test/implem/test_SimulationHistory.cu:12-77: construction,
// per-iteration setters/getters, resize, bulk set/get).  Prints "ok" and returns 0, or names the failed check.
#include <array>
#include <cstdio>
#include <vector>

#include "core/SimulationHistory.hpp"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

int main()
{
    {
        SimulationHistory<float> h(10);
        CHECK(h.getNumIterations() == 10);
        h.setEnergyAt(0, 42.5f);
        CHECK(h.getEnergyAt(0) == 42.5f);
        h.setAngMomentumAt(0, 3.14f);
        CHECK(h.getAngMomentumAt(0) == 3.14f);
        h.setDensityCenterAt(0, {1.0f, 2.0f, 3.0f});
        const std::array<float, 3> c = h.getDensityCenterAt(0);
        CHECK(c[0] == 1.0f && c[1] == 2.0f && c[2] == 3.0f);
        h.setNumIterations(20);
        CHECK(h.getNumIterations() == 20);
        CHECK(h.getEnergyAt(0) == 42.5f);          // growing keeps what was recorded
        CHECK(h.getEnergyAt(19) == 0.0f);
    }
    {
        SimulationHistory<float> h(5);
        const std::vector<float> e = {1, 2, 3, 4, 5}, l = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f};
        std::vector<std::array<float, 3>> c;
        for (int k = 0; k < 5; ++k) c.push_back({3.0f * k + 1, 3.0f * k + 2, 3.0f * k + 3});
        h.setAllEnergy(e);
        h.setAllAngMomentum(l);
        h.setAllDensityCenter(c);
        CHECK(h.getAllEnergy() == e);
        CHECK(h.getAllAngMomentum() == l);
        CHECK(h.getAllDensityCenter() == c);
        CHECK(h.getNumIterations() == 5);
    }
    {
        SimulationHistory<double> h;                // default: empty, grows on demand
        CHECK(h.getNumIterations() == 0);
        h.setNumIterations(-3);
        CHECK(h.getNumIterations() == 0);
        bool threw = false;
        try { h.setEnergyAt(0, 1.0); } catch (const std::out_of_range&) { threw = true; }
        CHECK(threw);                               // out-of-range rows are an error, not a silent write
        threw = false;
        try { h.saveMetricsToCSV("/nonexistent_dir_for_murb_test/metrics.csv"); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    }
    if (failures == 0) std::printf("ok\n");
    return failures == 0 ? 0 : 1;
}
