// tsf_pool.h -- the library's host threads: ONE pool per process, started at the first parallel phase and kept.
//
// Round 6 (measured on the 256-thread GPU box, profiles/r06_host/): the host stages either side of the kernels -- the
// directory walk, the reader's load / count / parse phases, the packer's passes, the blob writer, the forecast sink --
// each started and joined their own 32 std::threads, ten to fifteen times per job: a few hundred thread creations per
// 10 000 series, 1-3 ms per phase whatever its size (a 2 500-directory walk cost as much as a 10 000-directory one), and
// more threads only made it worse (reader: 77 ms at 32 threads, 187 ms at 256).  A phase is now `run(k, f)`: k - 1
// tickets on the pool's queue, index 0 on the calling thread, which then takes whatever indices nobody has claimed yet
// (a busy pool never blocks a caller: it degrades to the caller's own thread) and waits for the claimed ones.  Several
// callers may run phases at the same time (the jobs' pipeline stages do).
#ifndef TSF_POOL_H
#define TSF_POOL_H

#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace tsfpool {

struct Job {
    std::function<void(int)> f;
    int k = 0;
    std::atomic<int> next{0}, done{0}, failed{0};
    std::mutex m;
    std::condition_variable cv;
    void run_claimed() {            // claim indices until none is left
        for (;;) {
            const int w = next.fetch_add(1);
            if (w >= k) return;
            try {
                f(w);
            } catch (...) {
                failed.store(1);
            }
            if (done.fetch_add(1) + 1 == k) {
                std::lock_guard<std::mutex> lk(m);
                cv.notify_all();
            }
        }
    }
};

class Pool {
 public:
    static Pool &get() {
        Pool *p = instance().load(std::memory_order_acquire);
        if (p) return *p;
        static std::mutex mk;
        std::lock_guard<std::mutex> lk(mk);
        p = instance().load(std::memory_order_acquire);
        if (!p) {
            p = new Pool();
            instance().store(p, std::memory_order_release);
            static bool hooked = false;
            if (!hooked) { pthread_atfork(nullptr, nullptr, &Pool::after_fork_child); hooked = true; }
        }
        return *p;
    }
    int workers() const { return (int)th_.size(); }
    // f(w) for w in [0, k): returns when every index has run.  Throws std::bad_alloc if an f(w) threw.
    template <class F>
    void run(int k, F &&f) {
        if (k <= 1) {
            if (k == 1) f(0);
            return;
        }
        auto job = std::make_shared<Job>();
        job->f = std::forward<F>(f);
        job->k = k;
        {
            std::lock_guard<std::mutex> lk(m_);
            const int tickets = k - 1 < (int)th_.size() ? k - 1 : (int)th_.size();
            for (int i = 0; i < tickets; ++i) q_.push_back(job);
        }
        cv_.notify_all();
        job->run_claimed();
        {
            std::unique_lock<std::mutex> lk(job->m);
            job->cv.wait(lk, [&] { return job->done.load() >= job->k; });
        }
        if (job->failed.load()) throw std::bad_alloc();
    }

 private:
    Pool() {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        int n = hw < 32 ? hw : 32;
        if (const char *e = std::getenv("TSF_POOL_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) n = v; }
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
        for (auto &t : th_) t.detach();          // (process lifetime: nothing to join at exit, nothing to inherit across a fork)
    }
    void loop() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty(); });
                job = std::move(q_.front());
                q_.pop_front();
            }
            job->run_claimed();
        }
    }
    static std::atomic<Pool *> &instance() { static std::atomic<Pool *> p{nullptr}; return p; }
    // the child of a fork has none of the parent's threads: it starts a pool of its own at its first phase (the parent's
    // object, whose mutexes may be held by threads that do not exist here, is abandoned)
    static void after_fork_child() { instance().store(nullptr, std::memory_order_release); }

    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> q_;
    std::vector<std::thread> th_;
};

// f(begin, end, t) over n items in `parts` static contiguous pieces (piece t on whichever thread claims it)
template <class F>
inline void parallel_for(int64_t n, int parts, F f) {
    if (parts <= 1 || n <= 0) {
        f((int64_t)0, n, 0);
        return;
    }
    const int64_t per = (n + parts - 1) / parts;
    const int k = (int)((n + per - 1) / per);
    Pool::get().run(k, [&](int t) {
        const int64_t a = per * t, b = a + per < n ? a + per : n;
        if (a < b) f(a, b, t);
    });
}

}  // namespace tsfpool

#endif  // TSF_POOL_H
