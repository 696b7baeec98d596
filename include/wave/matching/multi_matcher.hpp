// wave::MultiMatcher<MatcherT, ParamsT>: a fixed crew of worker threads registering queued
// (ref, target) pairs concurrently.
//
// On the MI355X back end a worker is a host thread feeding the GPU through its OWN matcher (its
// own wm_ctx and HIP stream), so registrations of different workers overlap on the device.  For small
// clouds that alone leaves the chip idle between tiny kernels (10k-point pairs: ~5 000
// registrations/s however many workers), so a worker whose matcher can register many pairs in one
// launch (ICPMatcher::matchBatch: full-resolution targets up to 50 000 points, voxel-filtered matchers
// on scans up to 200 000) takes
// EVERYTHING that is queued -- up to 256 pairs, one compute unit each -- per trip: 80 000
// registrations/s with one worker, 110 000-145 000 with two to four at 10 000 points, 19 000-37 000 at
// 30 000 (libwave_amd/host/bench_multimatcher).  A batch is gathered over several refills of the queue
// (see work()): the reference's default queue of 10 feeds it as well as a deep one.
//
// Public surface as in the reference (wave_matching/include/wave/matching/multi_matcher.hpp:
// 29-96): construct with (n_threads, queue_size, params); insert(id, ref, target) blocks while
// the job queue is full; done() tells whether every inserted pair has been registered;
// getResult(&id, &T, &info) -- declared there (multi_matcher.hpp:64-77) but never defined -- pops one
// finished job, BLOCKING while none is finished but pairs are still pending, as its comment there says.
// The machinery underneath is this file's own: one mutex, four condition variables (idle workers, the
// gatherer, insert(), getResult()), a gather protocol for batches, plain structs.
#ifndef WAVE_MULTI_MATCHER_HPP
#define WAVE_MULTI_MATCHER_HPP

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "wave/matching/matcher.hpp"
#include "wave/matching/pcl_common.hpp"
#include "wave/utils/math.hpp"

#if defined(__SANITIZE_THREAD__)
#define WAVE_MATCHING_TSAN 1
#elif defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define WAVE_MATCHING_TSAN 1
#endif
#endif
#ifndef WAVE_MATCHING_TSAN
#define WAVE_MATCHING_TSAN 0
#endif

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
        const std::vector<int> devs = devices();
        // workers are dealt onto the device SLOTS round robin; the batched path's bookkeeping (which workers
        // gather batches, how many batches are in flight) is per slot, so that every listed GPU gets batches
        const size_t slots = devs.empty() ? 1 : devs.size();
        batches_in_flight_.assign(slots, 0);
        idle_batch_workers_.assign(slots, 0);
        batches_run_.assign(slots, 0);
        for (int w = 0; w < crew; ++w) {
            const int slot = (int) ((size_t) w % slots);
            const int dev = devs.empty() ? -1 : devs[(size_t) slot];
            const int rank = (int) ((size_t) w / slots);  // this worker's number among its slot's workers
            workers_.emplace_back([this, dev, slot, rank] { work(dev, slot, rank); });
        }
    }

    ~MultiMatcher() {
        {
            std::lock_guard<std::mutex> hold(lock_);
            closing_ = true;
        }
        jobs_changed_.notify_all();
        more_jobs_.notify_all();
        space_free_.notify_all();
        results_changed_.notify_all();
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
    /** HIP devices the workers of pools constructed afterwards are dealt onto, round robin (one pool
     *  feeding all GPUs of a node: registrations are independent, so throughput scales with the
     *  devices -- the "replicas" way of using several GPUs; ONE registration over several GPUs is
     *  ICPMatcher::setDevices).  Empty: every worker uses the default device.  Env
     *  WAVE_MATCHING_DEVICES="0,1,2,3" does the same.  Needs a matcher type with a static
     *  setThreadDevice(int) (ICPMatcher, GICPMatcher, NDTMatcher). */
    static void setDevices(const std::vector<int> &list) { deviceSetting() = list; }
    static std::vector<int> devices() {
        if (!deviceSetting().empty()) return deviceSetting();
        std::vector<int> out;
        if (const char *e = std::getenv("WAVE_MATCHING_DEVICES")) {
            for (const char *p = e; *p;) {
                char *end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) break;
                if (v >= 0) out.push_back((int) v);
                p = *end == ',' ? end + 1 : end;
            }
        }
        return out;
    }
    /** Number of worker threads of this pool. */
    int workers() const { return static_cast<int>(workers_.size()); }
    /** Batched launches run so far by the workers of each device slot (slot k = the k-th entry of devices(); one
     *  slot without a device list).  Diagnostic: tells whether every GPU of the list is being fed batches. */
    std::vector<int> batchesPerSlot() {
        std::lock_guard<std::mutex> hold(lock_);
        return batches_run_;
    }

    MultiMatcher(const MultiMatcher &) = delete;
    MultiMatcher &operator=(const MultiMatcher &) = delete;

    /** Queues one pair; blocks while `queue_size` pairs are already waiting. */
    void insert(const int &id, const PCLPointCloudPtr &src, const PCLPointCloudPtr &target) {
        std::unique_lock<std::mutex> hold(lock_);
        space_free_.wait(hold, [this] { return jobs_.size() < capacity_; });
        jobs_.push_back(Job{id, src, target});
        ++unfinished_;
        const bool lingering = lingering_;
        hold.unlock();
        // the worker that is gathering a batch if there is one (the usual case while pairs keep coming: ONE
        // thread woken per pair, not the crew), else the idle workers
        if (lingering) more_jobs_.notify_one();
        else jobs_changed_.notify_all();  // (the workers' conditions differ -- batch workers, the bound: all look)
    }

    /** True once every inserted pair has been registered (successfully or not). */
    bool done() {
        std::lock_guard<std::mutex> hold(lock_);
        return unfinished_ == 0;
    }

    /** Pops the oldest finished registration.  Blocks until one is ready while the output is empty but
     *  inserted pairs are still pending; returns false only when the output is empty AND nothing is pending
     *  (the contract the reference documents, multi_matcher.hpp:64-77): `insert` N pairs, then
     *  `while (getResult(...))` collects exactly N. */
    bool getResult(int *id, Eigen::Affine3d *transform, Mat6 *info) {
        std::unique_lock<std::mutex> hold(lock_);
        results_changed_.wait(hold, [this] { return !finished_.empty() || unfinished_ == 0 || closing_; });
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

    // Matchers that can register many pairs in one device launch (ICPMatcher::matchBatch) get
    // everything that is waiting in the queue at once -- up to kBatch pairs: one compute unit each
    // on a 256-CU MI355X -- instead of one pair per trip.
    static constexpr size_t kBatch = 256;
    template <typename M>
    static auto takeBatch(M &matcher, std::deque<Job> &jobs, std::vector<Job> &taken, int)
        -> decltype(matcher.batchable(jobs.front().ref, jobs.front().target), void()) {
        while (!jobs.empty() && taken.size() < kBatch && matcher.batchable(jobs.front().ref, jobs.front().target)) {
            taken.push_back(jobs.front());
            jobs.pop_front();
        }
    }
    template <typename M>
    static void takeBatch(M &, std::deque<Job> &, std::vector<Job> &, long) {}
    // would the pair in front go down the batched path?
    template <typename M>
    static auto frontBatchable(M &matcher, const std::deque<Job> &jobs, int)
        -> decltype(matcher.batchable(jobs.front().ref, jobs.front().target), bool()) {
        return !jobs.empty() && matcher.batchable(jobs.front().ref, jobs.front().target);
    }
    template <typename M>
    static bool frontBatchable(M &, const std::deque<Job> &, long) {
        return false;
    }
    // Batches in flight at once (gathered, staged, launched, waited for -- each by its own worker): one batch
    // fills the device, a second and third overlap their host-side staging and uploads with it; beyond that
    // more only add threads copying at the same time (measured on an MI355X box with a 16-CPU quota, 10k-point
    // pairs, queue of 10: 95 000 registrations/s with 4 workers, 74 000 with 8, 48 000 with 16 before this
    // bound).  The other workers of a big crew stay asleep -- or register the pairs that do not batch.
    static int maxBatchesInFlight() {
        if (const char *e = std::getenv("WAVE_MATCHING_MAX_BATCHES")) {
            const int v = std::atoi(e);
            if (v > 0) return v;
        }
        return 4;
    }
    template <typename M>
    auto runBatch(M &matcher, const std::vector<Job> &taken, int)
        -> decltype(matcher.matchBatch(std::declval<const std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> &>(),
                                       std::declval<typename M::BatchOutcomes &>()),
                    bool()) {
        std::vector<std::pair<PCLPointCloudPtr, PCLPointCloudPtr>> pairs;
        pairs.reserve(taken.size());
        for (const Job &j : taken) pairs.emplace_back(j.ref, j.target);
        typename M::BatchOutcomes got;
        if (!matcher.matchBatch(pairs, got) || got.size() != taken.size()) return false;
        {
            std::lock_guard<std::mutex> hold(lock_);
            for (size_t k = 0; k < taken.size(); ++k) {
                Outcome out;
                out.id = taken[k].id;
                out.transform = got[k].transform;
                out.info = got[k].info;
                finished_.push_back(out);
                --unfinished_;
            }
        }
        results_changed_.notify_all();
        return true;
    }
    template <typename M>
    bool runBatch(M &, const std::vector<Job> &, long) {
        return false;
    }

    void registerOne(T &matcher, const Job &job) {
        matcher.setup(job.ref, job.target);
        matcher.match();
        matcher.estimateInfo();

        Outcome out;
        out.id = job.id;
        out.transform = matcher.getResult();
        out.info = matcher.getInfo();
        {
            std::lock_guard<std::mutex> hold(lock_);
            finished_.push_back(out);
            --unfinished_;
        }
        results_changed_.notify_all();  // (all: the last result also releases every getResult() that must return false)
    }

    // worker body: the matcher lives on this thread's stack, so its device context is created
    // (lazily, at the first match) by the thread that uses it
    template <typename M>
    static auto bindThread(int device, int) -> decltype(M::setThreadDevice(device), void()) {
        M::setThreadDevice(device);
    }
    template <typename M>
    static void bindThread(int, long) {}

    // How a batch forms.  The reference's default queue holds TEN pairs (multi_matcher.hpp:32-34), a launch of
    // the batched path wants a couple of hundred (one compute unit each, ~3 ms whatever their number).  So
    // the queue's capacity must not be the batch's: ONE worker at a time -- the gatherer -- takes what is
    // queued, frees the slots, and keeps taking while the producer keeps delivering, until it holds kBatch
    // pairs or nothing new has come for kLingerUs (a producer that paused: a lone pair is delayed by that
    // much, against the milliseconds it takes to register); then the next worker gathers while this one
    // stages and launches.  With a queue of 10 the pool now forms the same batches as with a queue of
    // thousands, and more workers only add staging threads: throughput no longer falls with the crew.
    static constexpr int kLingerUs = 30;
    // may a worker of device slot `slot` take the (batchable) pair in front?  Batches are gathered by the first
    // max_batches_ workers OF EACH SLOT (the same few contexts per GPU, their staging buffers warm), at most
    // max_batches_ in flight per slot.  Another worker steps in only when no batch worker anywhere could take
    // the pair (they are all busy registering pairs one at a time): no head-of-line blocking behind them.
    bool mayGather(int slot, bool batch_worker) const {
        if (batches_in_flight_[(size_t) slot] >= max_batches_) return false;
        if (batch_worker) return true;
        for (size_t s = 0; s < idle_batch_workers_.size(); ++s)
            if (idle_batch_workers_[s] > 0 && batches_in_flight_[s] < max_batches_) return false;
        return true;
    }
    void work(int device, int slot, int rank) {
        const bool batch_worker = rank < max_batches_;
        if (device >= 0) bindThread<T>(device, 0);
        T matcher{R(config_)};
        std::vector<Job> taken;
        for (;;) {
            Job job;
            taken.clear();
            {
                std::unique_lock<std::mutex> hold(lock_);
                if (batch_worker) ++idle_batch_workers_[(size_t) slot];
                jobs_changed_.wait(hold, [this, &matcher, batch_worker, slot] {
                    return closing_ || (!jobs_.empty() && !gathering_ &&
                                        (!frontBatchable(matcher, jobs_, 0) || mayGather(slot, batch_worker)));
                });
                if (batch_worker) --idle_batch_workers_[(size_t) slot];
                if (closing_) return;
                gathering_ = true;
                takeBatch(matcher, jobs_, taken, 0);
                if (taken.empty()) {  // the matcher has no batch path, or this pair does not fit it: one at a time
                    job = jobs_.front();
                    jobs_.pop_front();
                } else {
                    while (taken.size() < kBatch && !closing_) {
                        if (jobs_.empty()) {
                            space_free_.notify_all();  // (the producer may be waiting for the slots just freed)
                            lingering_ = true;
                            // The monotonic clock: a wall-clock step backwards must not stretch the linger (the
                            // gatherer holds `gathering_` meanwhile).  Under ThreadSanitizer only, the system clock:
                            // wait_for is pthread_cond_clockwait, which GCC 11's TSan does not know -- it misses the
                            // unlock inside the wait and reports races that are not there.
#if WAVE_MATCHING_TSAN
                            const bool more = more_jobs_.wait_until(hold, std::chrono::system_clock::now() + std::chrono::microseconds(kLingerUs),
                                                                    [this] { return closing_ || !jobs_.empty(); });
#else
                            const bool more = more_jobs_.wait_for(hold, std::chrono::microseconds(kLingerUs),
                                                                  [this] { return closing_ || !jobs_.empty(); });
#endif
                            lingering_ = false;
                            if (!more || closing_) break;
                        }
                        const size_t before = taken.size();
                        takeBatch(matcher, jobs_, taken, 0);
                        if (taken.size() == before) break;  // (the pair in front is for the one-at-a-time path)
                    }
                    if (taken.size() < 2) {  // nothing to gain from a launch of one
                        job = taken.front();
                        taken.clear();
                    } else {
                        ++batches_in_flight_[(size_t) slot];
                        ++batches_run_[(size_t) slot];
                    }
                }
                gathering_ = false;
            }
            space_free_.notify_all();    // slots are free for insert() ...
            jobs_changed_.notify_all();  // ... and the next worker may gather
            if (!taken.empty()) {
                const bool ran = runBatch(matcher, taken, 0);
                {
                    std::lock_guard<std::mutex> hold(lock_);
                    --batches_in_flight_[(size_t) slot];
                }
                jobs_changed_.notify_all();  // (a worker held back by the bound may gather now)
                if (!ran)
                    for (const Job &j : taken) registerOne(matcher, j);  // (the device refused: one by one)
                continue;
            }

            registerOne(matcher, job);
        }
    }

    static std::vector<int> &deviceSetting() {
        static std::vector<int> setting;
        return setting;
    }
    static int &maxWorkersSetting() {
        static int setting = 0;
        return setting;
    }

    const size_t capacity_;
    R config_;
    std::mutex lock_;
    std::condition_variable jobs_changed_;  // idle workers: a pair is waiting and nobody is gathering
    std::condition_variable more_jobs_;     // the gatherer, between the producer's deliveries
    std::condition_variable space_free_;    // insert(): the queue has room
    std::condition_variable results_changed_;  // getResult(): a registration finished (or nothing is pending any more)
    bool gathering_ = false, lingering_ = false;
    std::vector<int> batches_in_flight_;    // per device slot
    std::vector<int> idle_batch_workers_;   // per device slot: batch workers waiting for work
    std::vector<int> batches_run_;          // per device slot: batches launched so far (batchesPerSlot)
    const int max_batches_ = maxBatchesInFlight();
    std::deque<Job> jobs_;
    std::deque<Outcome, Eigen::aligned_allocator<Outcome>> finished_;
    size_t unfinished_ = 0;
    bool closing_ = false;
    std::vector<std::thread> workers_;
};

}  // namespace wave

#endif  // WAVE_MULTI_MATCHER_HPP
