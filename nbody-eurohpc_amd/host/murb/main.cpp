// murb-hip: a murb-compatible driver for the MI355X implementations.  The reference's own
// src/murb/main.cpp cannot be built without MPI/OpenGL, so this restates the part of it that the
// benchmark command lines use (README.md:54-102): same flags, same configuration banner, same
// iteration loop and timing window, same final line
//     Entire simulation took <ms> ms (<fps> FPS, <gflops> Gflop/s)
// (reference main.cpp:61-165 flags, :205-270 factory, :323-334 banner, :348-398 loop and summary).
//
// A maintainer of the reference adds the HIP path to the real driver with one branch in
// createImplem() — see INTEGRATION.md; the branch below is that code.
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "core/Bodies.hpp"
#include "core/BodiesAllocator.hpp"
#include "core/SimulationHistory.hpp"
#include "implem/SimulationNBodyHIP.hpp"
#include "implem/SimulationNBodyHIPTracking.hpp"
#include "murbhip.h"
#include "utils/ArgumentsReader.hpp"
#include "utils/Perf.hpp"

// same globals and defaults as the reference (main.cpp:38-52)
unsigned long NBodies;
unsigned long NIterations;
std::string ImplTag = "hip+tile";
bool Verbose = false;
bool GSEnable = true;
bool VisuEnable = true;
float Dt = 3600;
float Softening = 2e+08;
std::string BodiesScheme = "galaxy";
bool ShowGFlops = false;
int NDevices = 0;        // --ngpu, hip+tile+multi only (0 = all visible)
bool FreeRunning = false;   // --free: sync once at the end instead of once per iteration
bool DeviceInit = false;    // --dinit: generate the initial conditions on the device (bit-identical to the host's)
std::string MetricsFile;    // --csv: where hip+tracking / hip+leapfrog save their history
std::shared_ptr<SimulationHistory<double>> History;

// One row per command-line option: tag (as Arguments_reader wants it: "-im" is typed "--im"), name of
// its value ("" = flag), required?, help text.  Same options as the reference (main.cpp:66-112) minus
// OpenCL's, plus --ngpu and --free.
struct Option {
    const char *tag, *value;
    bool required;
    std::string help;
};

static std::vector<Option> optionTable()
{
    return {
        {"n", "nBodies", true, "the number of generated bodies."},
        {"i", "nIterations", true, "the number of iterations to compute."},
        {"v", "", false, "enable verbose mode."},
        {"h", "", false, "display this help."},
        {"-help", "", false, "display this help."},
        {"-dt", "timeStep", false, "select a fixed time step in second (default is " + std::to_string(Dt) + " sec)."},
        {"-ngs", "", false, "accepted for compatibility (no visualization in this driver)."},
        {"-nv", "", false, "no visualization (always the case here)."},
        {"-nvc", "", false, "accepted for compatibility."},
        {"-ww", "winWidth", false, "accepted for compatibility."},
        {"-wh", "winHeight", false, "accepted for compatibility."},
        {"-im", "ImplTag", false,
         "code implementation tag:\n"
         "\t\t\t - \"hip+tile\"        one MI355X, device-resident bodies\n"
         "\t\t\t - \"hip+tile+multi\"  bodies partitioned over --ngpu MI355X, RCCL position exchange\n"
         "\t\t\t - \"hip+tracking\"    hip+tile + energy / angular momentum / centre of mass per iteration\n"
         "\t\t\t - \"hip+leapfrog\"    hip+tracking with a kick-drift-kick leapfrog integrator\n"
         "\t\t\t ----"},
        {"-soft", "softeningFactor", false, "softening factor."},
        {"s", "bodies scheme", false, "bodies scheme (initial conditions can be \"galaxy\" or \"random\")."},
        {"-gf", "", false, "display the number of GFlop/s."},
        {"-ngpu", "nGpus", false, "number of GPUs for hip+tile+multi (default: all visible)."},
        {"-free", "", false, "free-running timing: one device sync at the end, not one per iteration."},
        {"-dinit", "", false, "generate the initial conditions on the device (same bodies, bit for bit)."},
        {"-csv", "file", false, "hip+tracking / hip+leapfrog: save the metrics history as CSV."},
    };
}

static void argsReader(int argc, char **argv)
{
    std::map<std::string, std::string> reqArgs, faculArgs, docArgs;
    for (const Option &o : optionTable()) {
        (o.required ? reqArgs : faculArgs)[o.tag] = o.value;
        docArgs[o.tag] = o.help;
    }
    Arguments_reader reader(argc, argv);
    const bool complete = reader.parse_arguments(reqArgs, faculArgs);
    if (!complete || reader.exist_argument("h") || reader.exist_argument("-help")) {
        if (reader.parse_doc_args(docArgs)) reader.print_usage();
        else std::cout << "A problem was encountered when parsing arguments documentation... exiting." << std::endl;
        exit(-1);
    }
    const auto given = [&](const char *tag) { return reader.exist_argument(tag); };
    NBodies = stoi(reader.get_argument("n"));
    NIterations = stoi(reader.get_argument("i"));
    Verbose = given("v");
    GSEnable = !given("-ngs");
    VisuEnable = !given("-nv");
    ShowGFlops = given("-gf");
    FreeRunning = given("-free");
    DeviceInit = given("-dinit");
    if (given("-dt")) Dt = stof(reader.get_argument("-dt"));
    if (given("-im")) ImplTag = reader.get_argument("-im");
    if (given("s")) BodiesScheme = reader.get_argument("s");
    if (given("-ngpu")) NDevices = stoi(reader.get_argument("-ngpu"));
    if (given("-csv")) MetricsFile = reader.get_argument("-csv");
    if (given("-soft")) {
        Softening = stof(reader.get_argument("-soft"));
        if (Softening == 0.f) {   // the reference refuses it too (main.cpp:147-150): the self term would be 0/0
            std::cout << "Softening factor can't be equal to 0... exiting." << std::endl;
            exit(-1);
        }
    }
}

// "..d ..h ..m ..s" (main.cpp:175-196)
static std::string strDate(float timestamp)
{
    const unsigned days = timestamp / 86400.f;
    float rest = timestamp - days * 86400.f;
    const unsigned hours = rest / 3600.f;
    rest -= hours * 3600.f;
    const unsigned minutes = rest / 60.f;
    rest -= minutes * 60.f;
    std::stringstream res;
    res << std::fixed << std::setprecision(0) << std::setw(4) << days << "d " << std::setw(4) << hours << "h "
        << std::setw(4) << minutes << "m " << std::setprecision(3) << std::setw(5) << rest << "s";
    return res.str();
}

template <typename T> static SimulationNBodyHIP<T> *createImplem()
{
    if (ImplTag == "hip+tile") {
        HIPBodiesAllocator<T> hipAllocator(NBodies, BodiesScheme);
        return new SimulationNBodyHIP<T>(hipAllocator, Softening);
    }
    if (ImplTag == "hip+tile+multi") {
        int visible = 0;
        murbhipCheck(murbhip_device_count(&visible), "murbhip_device_count");
        const int use = NDevices > 0 ? NDevices : visible;
        std::vector<int> devices(use);
        for (int d = 0; d < use; ++d) devices[d] = d % (visible > 0 ? visible : 1);
        HIPBodiesAllocator<T> hipAllocator(NBodies, BodiesScheme);
        return new SimulationNBodyHIP<T>(hipAllocator, Softening, devices, /*exchange: RCCL when distinct GPUs*/
                                         use <= visible ? 1 : 0);
    }
    if (ImplTag == "hip+tracking" || ImplTag == "hip+leapfrog") {   // shaped like main.cpp:245-261
        HIPBodiesAllocator<T> hipAllocator(NBodies, BodiesScheme);
        History = std::make_shared<SimulationHistory<double>>((int)NIterations);
        return new SimulationNBodyHIPTracking<T, double>(hipAllocator, History, Softening, ImplTag == "hip+leapfrog");
    }
    std::cout << "Implementation '" << ImplTag << "' does not exist... Exiting." << std::endl;
    exit(-1);
}

int main(int argc, char **argv)
{
    argsReader(argc, argv);
    SimulationNBodyHIP<float> *simu = createImplem<float>();
    if (DeviceInit) {
        if (BodiesScheme != "galaxy" && BodiesScheme != "random") {
            std::cout << "--dinit supports the \"galaxy\" and \"random\" schemes... exiting." << std::endl;
            exit(-1);
        }
        std::dynamic_pointer_cast<HIPBodies<float>>(simu->getBodies())->initOnDevice(BodiesScheme, 0);
    }
    NBodies = simu->getBodies()->getN();
    const float Mbytes = simu->getAllocatedBytes() / 1024.f / 1024.f;

    std::cout << "n-body simulation configuration:" << std::endl;
    std::cout << "--------------------------------" << std::endl;
    std::cout << "  -> bodies scheme     (-s    ): " << BodiesScheme << std::endl;
    std::cout << "  -> implementation    (--im  ): " << ImplTag << std::endl;
    std::cout << "  -> nb. of bodies     (-n    ): " << NBodies << std::endl;
    std::cout << "  -> nb. of iterations (-i    ): " << NIterations << std::endl;
    std::cout << "  -> verbose mode      (-v    ): " << ((Verbose) ? "enable" : "disable") << std::endl;
    std::cout << "  -> precision                 : " << "fp32" << std::endl;
    std::cout << "  -> mem. allocated            : " << Mbytes << " MB" << std::endl;
    std::cout << "  -> geometry shader   (--ngs ): " << ((GSEnable) ? "enable" : "disable") << std::endl;
    std::cout << "  -> time step         (--dt  ): " << std::to_string(Dt) + " sec" << std::endl;
    std::cout << "  -> softening factor  (--soft): " << Softening << std::endl;

    simu->setDt(Dt);
    std::cout << "Simulation started..." << std::endl;

    // Timed region = computeOneIteration() + the driver's device sync, once per iteration
    // (main.cpp:353-371).  --free moves the sync after the loop (not a mode of the reference).
    Perf perfIte, perfTotal;
    float physicTime = 0.f;
    unsigned long iIte;
    if (FreeRunning) perfIte.start();
    for (iIte = 1; iIte <= NIterations; iIte++) {
        if (!FreeRunning) perfIte.start();
        simu->computeOneIteration();
        if (!FreeRunning) {
            simu->synchronize();
            perfIte.stop();
            perfTotal += perfIte;
        }
        physicTime += simu->getDt();
        if (Verbose && !FreeRunning) {
            std::stringstream gflops;
            if (ShowGFlops)
                gflops << ", " << std::setprecision(1) << std::fixed << std::setw(6)
                       << perfTotal.getGflops(simu->getFlopsPerIte() * iIte) << " Gflop/s";
            std::cout << "Iteration n°" << std::setw(4) << iIte << " (" << std::setprecision(1) << std::fixed
                      << std::setw(6) << perfTotal.getFPS(iIte) << " FPS" << gflops.str()
                      << "), physic time: " << strDate(physicTime) << "\r";
            if (iIte % 5 == 0) std::cout << std::flush;
        }
    }
    if (FreeRunning) {
        simu->synchronize();
        perfIte.stop();
        perfTotal += perfIte;
    }
    if (Verbose) std::cout << std::endl;
    std::cout << "Simulation ended." << std::endl << std::endl;

    std::stringstream gflops;
    if (ShowGFlops)
        gflops << ", " << std::setprecision(1) << std::fixed << std::setw(6)
               << perfTotal.getGflops(simu->getFlopsPerIte() * (iIte - 1)) << " Gflop/s";
    std::cout << "Entire simulation took " << perfTotal.getElapsedTime() << " ms "
              << "(" << perfTotal.getFPS(iIte - 1) << " FPS" << gflops.str() << ")" << std::endl;

    if (History && History->getNumIterations() > 1) {
        const double e0 = History->getEnergyAt(0), e1 = History->getEnergyAt(History->getNumIterations() - 1);
        std::cout << "Energy at the first / last tracked iteration: " << std::setprecision(9) << e0 << " / " << e1
                  << " J (relative drift " << std::setprecision(3) << (e1 - e0) / (e0 != 0 ? std::abs(e0) : 1.0) << ")"
                  << std::endl;
    }
    if (History && !MetricsFile.empty()) {
        // the reference keeps this export commented out (main.cpp:400-401: "metrics.csv")
        History->saveMetricsToCSV(MetricsFile);
        std::cout << "Metrics saved to " << MetricsFile << std::endl;
    }

    delete simu;
    return EXIT_SUCCESS;
}
