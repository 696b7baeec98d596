// wave::MultiMatcher<MatcherT, ParamsT>: a fixed crew of worker threads registering queued
// (ref, target) pairs concurrently.
//
// On the MI355X back end concurrency is what fills the GPU when the clouds are small: every
// worker builds its OWN matcher inside its thread, i.e. its own wm_ctx with its own HIP stream,
// so kernels of different registrations overlap on the device (scripts/bench_multimatcher.py:
// 10k-point pairs go from ~1 900 to ~5 600 registrations/s with 8 workers).
//
// Public surface as in the reference (wave_matching/include/wave/matching/multi_matcher.hpp:
// 29-96): construct with (n_threads, queue_size, params); insert(id, ref, target) blocks while
// the job queue is full; done() tells whether every inserted pair has been registered;
// getResult(&id, &T, &info) -- declared there but never defined -- pops one finished job.
// The machinery underneath is this file's own: one mutex, two condition variables, plain structs.
#ifndef WAVE_MULTI_MATCHER_HPP
#define WAVE_MULTI_MATCHER_HPP

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"
#include "wave/utils/math.hpp"

namespace wave {

template <typename T, typename R>
class MultiMatcher {
 public:
    MultiMatcher(int n_threads = std::thread::hardware_concurrency(), int queue_s = 10,
                 R params = R())
        : capacity_(queue_s > 0 ? static_cast<size_t>(queue_s) : 1), config_(params) {
        // The reference's default is one worker per hardware thread (each runs single-threaded PCL).
        // Here a worker is a host thread feeding ONE GPU: a few fill it (16 workers: 7 600
        // registrations/s at 10k points against 2 300 with one), and hundreds -- 256 on an MI355X
        // host -- only fight over the cores while they wait.  The crew is capped (default 16; env
        // WAVE_MATCHING_MAX_WORKERS or setMaxWorkers() to change it).
        const int crew = std::max(1, std::min(n_threads > 0 ? n_threads : 1, maxWorkers()));
        workers_.reserve(crew);
        for (int w = 0; w < crew; ++w) workers_.emplace_back([this] { work(); });
    }

    ~MultiMatcher() {
        {
            std::lock_guard<std::mutex> hold(lock_);
            closing_ = true;
        }
        jobs_changed_.notify_all();
        for (auto &w : workers_) w.join();
    }

    /** Upper bound on the worker crew of matchers constructed afterwards (<= 0: back to the default). */
    static void setMaxWorkers(int n) { maxWorkersSetting() = n; }
    static int maxWorkers() {
        if (maxWorkersSetting() > 0) return maxWorkersSetting();
        if (const char *e = std::getenv("WAVE_MATCHING_MAX_WORKERS")) {
            const int v = std::atoi(e);
            if (v > 0) return v;
        }
        return 16;
    }
    /** Number of worker threads of this pool. */
    int workers() const { return static_cast<int>(workers_.size()); }

    MultiMatcher(const MultiMatcher &) = delete;
    MultiMatcher &operator=(const MultiMatcher &) = delete;

    /** Queues one pair; blocks while `queue_size` pairs are already waiting. */
    void insert(const int &id, const PCLPointCloudPtr &src, const PCLPointCloudPtr &target) {
        std::unique_lock<std::mutex> hold(lock_);
        jobs_changed_.wait(hold, [this] { return jobs_.size() < capacity_; });
        jobs_.push_back(Job{id, src, target});
        ++unfinished_;
        hold.unlock();
        jobs_changed_.notify_all();
    }

    /** True once every inserted pair has been registered (successfully or not). */
    bool done() {
        std::lock_guard<std::mutex> hold(lock_);
        return unfinished_ == 0;
    }

    /** Pops the oldest finished registration; false when none is waiting. */
    bool getResult(int *id, Eigen::Affine3d *transform, Mat6 *info) {
        std::lock_guard<std::mutex> hold(lock_);
        if (finished_.empty()) return false;
        const Outcome &o = finished_.front();
        if (id) *id = o.id;
        if (transform) *transform = o.transform;
        if (info) *info = o.info;
        finished_.pop_front();
        return true;
    }

 private:
    struct Job {
        int id;
        PCLPointCloudPtr ref, target;
    };
    struct Outcome {
        EIGEN_MAKE_ALIGNED_OPERATOR_NEW
        int id;
        Eigen::Affine3d transform;
        Mat6 info;
    };

    // worker body: the matcher lives on this thread's stack, so its device context is created
    // (lazily, at the first match) by the thread that uses it
    void work() {
        T matcher{R(config_)};
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> hold(lock_);
                jobs_changed_.wait(hold, [this] { return closing_ || !jobs_.empty(); });
                if (closing_) return;
                job = jobs_.front();
                jobs_.pop_front();
            }
            jobs_changed_.notify_all();  // a slot is free for insert()

            matcher.setup(job.ref, job.target);
            matcher.match();
            matcher.estimateInfo();

            Outcome out;
            out.id = job.id;
            out.transform = matcher.getResult();
            out.info = matcher.getInfo();
            std::lock_guard<std::mutex> hold(lock_);
            finished_.push_back(out);
            --unfinished_;
        }
    }

    static int &maxWorkersSetting() {
        static int setting = 0;
        return setting;
    }

    const size_t capacity_;
    R config_;
    std::mutex lock_;
    std::condition_variable jobs_changed_;
    std::deque<Job> jobs_;
    std::deque<Outcome, Eigen::aligned_allocator<Outcome>> finished_;
    size_t unfinished_ = 0;
    bool closing_ = false;
    std::vector<std::thread> workers_;
};

}  // namespace wave

#endif  // WAVE_MULTI_MATCHER_HPP
