// TEST INFRASTRUCTURE: drives csrc/murb_crew.h (the per-shard host threads of libmurbhip.so) on the CPU — no HIP involved.
//   crew_selftest [members] [rounds]      prints "ok" and exits 0, or says what broke
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "murb_crew.h"

#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int members = argc > 1 ? std::atoi(argv[1]) : 8, rounds = argc > 2 ? std::atoi(argv[2]) : 3000;
    // 1. every member runs every job exactly once, on a thread of its own that on_start saw first
    {
        std::vector<std::atomic<int>> started(members), ran(members);
        std::vector<std::thread::id> ids(members);
        ShardCrew crew(members, [&](int i) { started[i].fetch_add(1); ids[i] = std::this_thread::get_id(); });
        CHECK(crew.threads() == (members > 1 ? members : 0), "thread count %d", crew.threads());
        for (int r = 0; r < rounds; ++r) {
            const int rc = crew.run([&](int i) { if (members > 1 && std::this_thread::get_id() != ids[i]) return -99; ran[i].fetch_add(1); return 0; });
            CHECK(rc == 0, "round %d returned %d", r, rc);
        }
        for (int i = 0; i < members; ++i) CHECK(ran[i].load() == rounds && (members == 1 || started[i].load() == 1), "member %d ran %d times", i, ran[i].load());
    }
    // 2. meet() is a barrier: what every member wrote before it is what every member reads after it, three phases per job,
    //    with members of very different speed; and run() is one too (a job never overlaps the next)
    {
        std::vector<std::atomic<long>> slot(members);
        std::atomic<int> bad{0};
        ShardCrew crew(members);
        for (int r = 0; r < rounds; ++r) {
            crew.run([&](int i) {
                for (int ph = 0; ph < 3; ++ph) {
                    const long stamp = (long)r * 3 + ph + 1;
                    if (i % 3 == 1 && (r & 63) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));   // a slow member
                    slot[i].store(stamp, std::memory_order_release);
                    crew.meet();
                    for (int k = 0; k < members; ++k) if (slot[k].load(std::memory_order_acquire) != stamp) bad.fetch_add(1);
                    crew.meet();   // nobody overwrites a slot before everybody has read it
                }
                return 0;
            });
        }
        CHECK(bad.load() == 0, "%d reads saw a member on the wrong side of the barrier", bad.load());
    }
    // 3. a failing member: its code comes back (the first in member order), the others finish, the barriers still pair up, and
    //    the crew goes on working afterwards
    {
        ShardCrew crew(members);
        std::atomic<int> done{0};
        int rc = crew.run([&](int i) { const int mine = (i == members - 1 || i == members / 2) ? -(100 + i) : 0; crew.meet(); done.fetch_add(1); crew.meet(); return mine; });
        CHECK(rc == (members > 1 ? -(100 + members / 2) : -100) && done.load() == members, "failure path: rc %d, %d members finished", rc, done.load());
        rc = crew.run([&](int) { crew.meet(); return 0; });
        CHECK(rc == 0, "the crew did not recover: %d", rc);
    }
    // 4. members that have gone to sleep (no job for a while) wake up for the next one; destruction joins sleeping members
    {
        ShardCrew crew(members);
        std::atomic<int> ran{0};
        for (int r = 0; r < 5; ++r) {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            crew.run([&](int) { ran.fetch_add(1); return 0; });
        }
        CHECK(ran.load() == 5 * members, "after idling: %d job runs", ran.load());
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    std::puts("ok");
    return 0;
}
