// wave::MultiMatcher<T, R> -- same interface as the reference's
// wave_matching/include/wave/matching/multi_matcher.hpp:29-96 (thread pool, bounded
// input queue, output queue, insert / done), plus the getResult() the reference
// declares (multi_matcher.hpp:73) but never defines.  Each worker owns one matcher,
// i.e. one wm_ctx with its own HIP stream, so registrations from different workers
// overlap on the GPU; workers are spread round-robin over `devices`.
#ifndef WAVE_MULTI_MATCHER_HPP
#define WAVE_MULTI_MATCHER_HPP

#include <condition_variable>
#include <mutex>
#include <queue>
#include <thread>
#include <tuple>
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
        : n_thread(n_threads), queue_size(queue_s), config(params) {
        this->stop = false;
        this->remaining_matches = 0;
        this->initPool(params);
    }

    ~MultiMatcher();

    /** inserts a pair of scans into the queue to be matched; blocks while the queue is full */
    void insert(const int &id, const PCLPointCloudPtr &src, const PCLPointCloudPtr &target);

    /** Checks to see if all submitted pairs have been matched */
    bool done();

    /** Pops one finished result. @returns false if the output buffer is empty */
    bool getResult(int *id, Eigen::Affine3d *transform, Mat6 *info);

 private:
    const int n_thread;
    const int queue_size;
    int remaining_matches;
    R config;
    std::queue<std::tuple<int, PCLPointCloudPtr, PCLPointCloudPtr>> input;
    std::queue<std::tuple<int, Eigen::Affine3d, Mat6>> output;
    std::vector<std::thread> pool;
    std::vector<T, Eigen::aligned_allocator<T>> matchers;

    std::mutex ip_mutex, op_mutex, cnt_mutex;
    std::condition_variable ip_condition;
    std::condition_variable op_condition;
    bool stop;

    void spin(int threadid);
    void initPool(R params);
};

}  // namespace wave

#include "wave/matching/impl/multi_matcher_impl.hpp"

#endif  // WAVE_MULTI_MATCHER_HPP
