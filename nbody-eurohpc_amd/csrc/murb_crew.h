// ShardCrew: one host thread per shard of a context that drives several GPUs from one process.
//
// The reference drives its simulation from ONE host thread (main.cpp:348-354) and so does every caller of this library.  A
// context with several shards nevertheless ENQUEUES a step from one thread per shard: a shard's share of a step is 25-45 HIP
// calls (5 launches, 2 collectives or W peer copies, the events and event waits between them), measured at 100-140 us per
// shard and step when one thread issues them for 8 shards in turn — 0.8-1.1 ms of host time per step against the 0.9 ms of GPU
// work a rank of 8 has at N = 200 000 (profiles/r03_host_enqueue.txt) — and no collective can complete before the LAST
// shard's call has been issued.
//
//   run(job)   every member runs job(its index); returns the first non-zero result (in member order) when ALL have returned.
//              A job only enqueues GPU work, so this never waits for a device.  With one member there is no thread at all:
//              the caller runs job(0).
//   meet()     barrier among the members, to be called from inside a job by EVERY member the same number of times (also by a
//              member whose own work has failed): needed only where a member's stream has to wait for an event another
//              member's thread records (peer-copy exchange).
//
// Members spin briefly for the next job (murbhip_steps issues one per step, microseconds apart) and then sleep on a condition
// variable (the murb loop syncs the device between steps).  No HIP in here: `on_start(i)` runs once on member i's thread
// (the library binds the thread to the shard's device there) — tests/helpers/crew_selftest.cpp drives it on the CPU.
#ifndef MURB_CREW_H_
#define MURB_CREW_H_

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class ShardCrew {
public:
    ShardCrew(int members, std::function<void(int)> on_start = nullptr) : n_(members), on_start_(std::move(on_start))
    {
        if (n_ < 2) return;
        rc_.assign((size_t)n_, 0);
        for (int i = 0; i < n_; ++i) threads_.emplace_back([this, i] { work(i); });
    }
    ~ShardCrew()
    {
        if (threads_.empty()) return;
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++generation_; }
        cv_work_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    ShardCrew(const ShardCrew&) = delete;
    ShardCrew& operator=(const ShardCrew&) = delete;

    int threads() const { return (int)threads_.size(); }

    int run(const std::function<int(int)>& job)
    {
        if (threads_.empty()) return job(0);
        job_ = &job;
        remaining_.store(n_, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(m_); ++generation_; }
        cv_work_.notify_all();
        for (int spins = 0; remaining_.load(std::memory_order_acquire) != 0;) {   // enqueueing takes ~50 us: spin first
            if (++spins < 20000) relax();
            else {
                std::unique_lock<std::mutex> lk(m_);
                cv_done_.wait(lk, [this] { return remaining_.load(std::memory_order_acquire) == 0; });
            }
        }
        job_ = nullptr;
        for (int rc : rc_) if (rc != 0) return rc;
        return 0;
    }

    void meet()
    {
        if (threads_.empty()) return;
        const unsigned my = phase_.load(std::memory_order_acquire);
        if (arrived_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            arrived_.store(0, std::memory_order_relaxed);
            phase_.fetch_add(1, std::memory_order_release);
            return;
        }
        for (int spins = 0; phase_.load(std::memory_order_acquire) == my;) {
            if (++spins < 4000) relax(); else std::this_thread::yield();
        }
    }

private:
    static void relax() { __builtin_ia32_pause(); }
    void work(int i)
    {
        if (on_start_) on_start_(i);
        unsigned long seen = 0;
        for (;;) {
            {
                int spins = 0;
                while (generation_.load(std::memory_order_acquire) == seen && ++spins < 4000) relax();
                if (generation_.load(std::memory_order_acquire) == seen) {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_work_.wait(lk, [&] { return generation_.load(std::memory_order_acquire) != seen; });
                }
            }
            seen = generation_.load(std::memory_order_acquire);
            if (stop_) return;
            rc_[(size_t)i] = (*job_)(i);
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                cv_done_.notify_one();
            }
        }
    }

    int n_;
    std::function<void(int)> on_start_;
    std::vector<std::thread> threads_;
    std::vector<int> rc_;
    const std::function<int(int)>* job_ = nullptr;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::atomic<unsigned long> generation_{0};
    std::atomic<int> remaining_{0};
    std::atomic<int> arrived_{0};
    std::atomic<unsigned> phase_{0};
    bool stop_ = false;
};

#endif
