// C ABI over the C++ host mirror, for the Python tests and bench.py (ctypes):
//   * the product's own initial conditions (so bench.py never needs the oracle for its inputs),
//   * the mirrored plugin classes driven exactly like the reference's tests drive the originals
//     (src/test/implem/test_SimulationNBody.cpp:28-71, test_CUDABodies.cpp:23-75).
#include <cstring>
#include <memory>
#include <string>

#include "core/Bodies.hpp"
#include "core/BodiesAllocator.hpp"
#include "core/SimulationHistory.hpp"
#include "implem/SimulationNBodyHIP.hpp"
#include "implem/SimulationNBodyHIPTracking.hpp"

namespace {
struct Sim {
    std::string scheme;
    SimulationNBodyHIP<float> *sim = nullptr;
    std::shared_ptr<SimulationHistory<double>> history;   // tracking sims only
};
void copyState(const dataSoA_t<float> &d, float *qx, float *qy, float *qz, float *vx, float *vy, float *vz, float *m,
               float *r)
{
    const size_t bytes = d.qx.size() * sizeof(float);
    const struct { float *dst; const std::vector<float> *src; } f[] = {
        {qx, &d.qx}, {qy, &d.qy}, {qz, &d.qz}, {vx, &d.vx}, {vy, &d.vy}, {vz, &d.vz}, {m, &d.m}, {r, &d.r}};
    for (const auto &e : f)
        if (e.dst) std::memcpy(e.dst, e.src->data(), bytes);
}
}  // namespace

extern "C" {

unsigned long murbhost_padding(unsigned long n, const char *scheme)
{
    return Bodies<float>(n, std::string(scheme)).getPadding();
}

// n + padding entries per array (NULL pointers are skipped)
void murbhost_init_bodies(unsigned long n, const char *scheme, unsigned long seed, float *qx, float *qy, float *qz,
                          float *vx, float *vy, float *vz, float *m, float *r)
{
    const Bodies<float> b(n, std::string(scheme), seed);
    copyState(b.getDataSoA(), qx, qy, qz, vx, vy, vz, m, r);
}

// host integrator alone (Bodies::updatePositionsAndVelocities), n entries out
void murbhost_integrate(unsigned long n, const char *scheme, const float *ax, const float *ay, const float *az, float dt,
                        int steps, int on_device, float *qx, float *qy, float *qz, float *vx, float *vy, float *vz)
{
    accSoA_t<float> acc;
    acc.ax.assign(ax, ax + n); acc.ay.assign(ay, ay + n); acc.az.assign(az, az + n);
    std::unique_ptr<Bodies<float>> b;
    if (on_device) {
        auto *hb = new HIPBodies<float>(n, std::string(scheme));
        hb->bindDevice(2e8f, 6.67384e-11f);
        b.reset(hb);
    } else b.reset(new Bodies<float>(n, std::string(scheme)));
    for (int s = 0; s < steps; ++s) b->updatePositionsAndVelocities(acc, dt);
    const auto &d = b->getDataSoA();
    const size_t bytes = n * sizeof(float);
    std::memcpy(qx, d.qx.data(), bytes); std::memcpy(qy, d.qy.data(), bytes); std::memcpy(qz, d.qz.data(), bytes);
    std::memcpy(vx, d.vx.data(), bytes); std::memcpy(vy, d.vy.data(), bytes); std::memcpy(vz, d.vz.data(), bytes);
}

void *murbhost_sim_create(unsigned long n, const char *scheme, float soft, float dt, int ndev, const int *devices,
                          int exchange)
{
    auto *h = new Sim;
    h->scheme = scheme;
    HIPBodiesAllocator<float> alloc(n, h->scheme);
    h->sim = new SimulationNBodyHIP<float>(alloc, soft, std::vector<int>(devices, devices + ndev), exchange);
    h->sim->setDt(dt);
    return h;
}
// --im hip+tracking (leapfrog = 0) / hip+leapfrog (1)
void *murbhost_tracking_create(unsigned long n, const char *scheme, float soft, float dt, int leapfrog, int ndev,
                               const int *devices, int exchange)
{
    auto *h = new Sim;
    h->scheme = scheme;
    h->history = std::make_shared<SimulationHistory<double>>();
    HIPBodiesAllocator<float> alloc(n, h->scheme);
    h->sim = new SimulationNBodyHIPTracking<float, double>(alloc, h->history, soft, leapfrog != 0,
                                                           std::vector<int>(devices, devices + ndev), exchange);
    h->sim->setDt(dt);
    return h;
}
int murbhost_history_rows(void *p)
{
    auto *h = static_cast<Sim *>(p);
    return h->history ? h->history->getNumIterations() : -1;
}
// rows entries each; centers as x0,y0,z0,x1,...
void murbhost_history_get(void *p, double *energy, double *ang_momentum, double *centers)
{
    const auto &hist = *static_cast<Sim *>(p)->history;
    for (int i = 0; i < hist.getNumIterations(); ++i) {
        energy[i] = hist.getEnergyAt(i);
        ang_momentum[i] = hist.getAngMomentumAt(i);
        for (int k = 0; k < 3; ++k) centers[3 * i + k] = hist.getDensityCenterAt(i)[k];
    }
}
// SimulationHistory alone (no device): fill `rows` rows and write the CSV.  0, or -1 if the file cannot be opened.
int murbhost_history_csv(const char *path, int rows, const double *energy, const double *ang_momentum, const double *centers)
{
    SimulationHistory<double> hist(rows);
    for (int i = 0; i < rows; ++i) {
        hist.setEnergyAt(i, energy[i]);
        hist.setAngMomentumAt(i, ang_momentum[i]);
        hist.setDensityCenterAt(i, {centers[3 * i], centers[3 * i + 1], centers[3 * i + 2]});
    }
    try {
        hist.saveMetricsToCSV(path);
    } catch (const std::runtime_error &) {
        return -1;
    }
    return 0;
}
int murbhost_sim_history_csv(void *p, const char *path)
{
    try {
        static_cast<Sim *>(p)->history->saveMetricsToCSV(path);
    } catch (const std::runtime_error &) {
        return -1;
    }
    return 0;
}

void murbhost_sim_destroy(void *p)
{
    auto *h = static_cast<Sim *>(p);
    if (h) { delete h->sim; delete h; }
}
void murbhost_sim_step(void *p, int iterations)
{
    auto *h = static_cast<Sim *>(p);
    for (int i = 0; i < iterations; ++i) h->sim->computeOneIteration();
    h->sim->synchronize();
}
// HIPBodies::initOnDevice with the scheme the simulation was created with
void murbhost_sim_init_on_device(void *p, unsigned long seed)
{
    auto *h = static_cast<Sim *>(p);
    std::dynamic_pointer_cast<HIPBodies<float>>(h->sim->getBodies())->initOnDevice(h->scheme, seed);
}
unsigned long murbhost_sim_n(void *p) { return static_cast<Sim *>(p)->sim->getBodies()->getN(); }
float murbhost_sim_flops_per_ite(void *p) { return static_cast<Sim *>(p)->sim->getFlopsPerIte(); }
float murbhost_sim_allocated_bytes(void *p) { return static_cast<Sim *>(p)->sim->getAllocatedBytes(); }
void murbhost_sim_state(void *p, float *qx, float *qy, float *qz, float *vx, float *vy, float *vz, float *m, float *r)
{
    copyState(static_cast<Sim *>(p)->sim->getBodies()->getDataSoA(), qx, qy, qz, vx, vy, vz, m, r);
}
void murbhost_sim_acc(void *p, float *ax, float *ay, float *az)
{
    const auto &a = static_cast<Sim *>(p)->sim->getAccSoA();
    const size_t bytes = a.ax.size() * sizeof(float);
    std::memcpy(ax, a.ax.data(), bytes); std::memcpy(ay, a.ay.data(), bytes); std::memcpy(az, a.az.data(), bytes);
}

}  // extern "C"
