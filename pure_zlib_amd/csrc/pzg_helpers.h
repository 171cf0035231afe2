// pzg_helpers.h -- the host-side helper thread pool of pzg_api.cpp (packing / copy-out of the host-pointer paths), in a header of
// its own so that the CPU suite can run it under ThreadSanitizer (tests/cxx/helpers_stress.cpp).  Plain C++17, no HIP.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// A pool of helper threads for the host-side packing / copy-out of the host-pointer paths.  The threads are started on first
// use and live as long as the context (round 4: creating two dozen threads per range cost more than the copying they did);
// run() may be called from several threads at once -- every job is a list of parts that the workers AND the caller take one
// at a time, so a job always completes even if no worker could be started.
class Helpers {
public:
    explicit Helpers(unsigned n) : n_(n ? n : 1u) {}
    ~Helpers()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    unsigned size() const { return n_; }
    // f(part, parts): every part in 0 .. parts - 1 runs exactly once, on some thread; returns when all are done
    template <class F>
    void run(unsigned parts, F &&f)
    {
        if (parts <= 1) {
            f(0u, 1u);
            return;
        }
        auto job = std::make_shared<Job>();
        job->parts = parts;
        job->fn = [&f, parts](unsigned p) { f(p, parts); };
        {
            std::lock_guard<std::mutex> g(mu_);
            start_workers();
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work_on(*job);
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [&] { return job->done == job->parts; });
        // the job leaves the queue with its caller (a worker may have taken it out already; with no worker at all -- none could
        // be started -- nobody else ever would, and the queue would grow by one job, holding a dangling reference, per call)
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
            if (*it == job) {
                jobs_.erase(it);
                break;
            }
    }

private:
    struct Job {
        std::function<void(unsigned)> fn;
        unsigned parts = 0;
        std::atomic<unsigned> next{0};
        unsigned done = 0;  // (under mu_)
    };
    void work_on(Job &j)
    {
        for (;;) {
            const unsigned p = j.next.fetch_add(1u);
            if (p >= j.parts) return;
            j.fn(p);
            bool last;
            {
                std::lock_guard<std::mutex> g(mu_);
                last = ++j.done == j.parts;
            }
            if (last) done_cv_.notify_all();
        }
    }
    void start_workers()  // (mu_ held)
    {
        while (workers_.size() + 1 < n_) {
            try {
                workers_.emplace_back([this] {
                    std::unique_lock<std::mutex> g(mu_);
                    for (;;) {
                        cv_.wait(g, [&] { return quit_ || !jobs_.empty(); });
                        if (quit_) return;
                        // the first job that still has a part to take (two callers -- a pack on the issuing thread, a copy-out on
                        // the draining one -- are served side by side; fully taken jobs in front leave the queue)
                        while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->parts) jobs_.pop_front();
                        if (jobs_.empty()) continue;
                        std::shared_ptr<Job> j = jobs_.front();
                        g.unlock();
                        work_on(*j);
                        g.lock();
                    }
                });
            } catch (...) {  // no more threads to be had: the callers do what the missing workers would have done
                break;
            }
        }
    }
    unsigned n_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::shared_ptr<Job>> jobs_;
    std::vector<std::thread> workers_;
    bool quit_ = false;
};

}  // namespace
