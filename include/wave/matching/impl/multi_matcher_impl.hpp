// Implementation of wave::MultiMatcher (reference:
// wave_matching/include/wave/matching/impl/multi_matcher_impl.hpp:9-93).
#ifndef WAVE_MULTI_MATCHER_IMPL_HPP
#define WAVE_MULTI_MATCHER_IMPL_HPP

namespace wave {

template <class T, class R>
MultiMatcher<T, R>::~MultiMatcher() {
    {
        // the reference flips `stop` without the lock (impl:11); taking it closes the
        // lost-wakeup window, behaviour is otherwise identical
        std::unique_lock<std::mutex> lock(this->ip_mutex);
        this->stop = true;
    }
    this->ip_condition.notify_all();
    for (int id = 0; id < this->n_thread; ++id) {
        this->pool.at(id).join();
    }
}

template <class T, class R>
void MultiMatcher<T, R>::initPool(R params) {
    this->config = params;
    this->matchers.reserve(this->n_thread);  // workers keep references into the vector
    for (int i = 0; i < this->n_thread; i++) {
        this->matchers.emplace_back(T(R(this->config)));
    }
    for (int i = 0; i < this->n_thread; i++) {
        this->pool.emplace_back(std::thread(&MultiMatcher<T, R>::spin, this, i));
    }
}

template <class T, class R>
void MultiMatcher<T, R>::spin(int threadid) {
    std::tuple<int, PCLPointCloudPtr, PCLPointCloudPtr> val;
    while (true) {
        {
            std::unique_lock<std::mutex> lock(this->ip_mutex);
            while (!this->stop && this->input.empty()) {
                this->ip_condition.wait(lock);
            }
            if (this->stop) {
                return;
            }
            val = this->input.front();
            this->input.pop();
            lock.unlock();
            this->ip_condition.notify_one();
        }
        this->matchers.at(threadid).setRef(std::get<1>(val));
        this->matchers.at(threadid).setTarget(std::get<2>(val));
        this->matchers.at(threadid).match();
        this->matchers.at(threadid).estimateInfo();
        {
            std::unique_lock<std::mutex> lockop(this->op_mutex);
            this->output.emplace(std::get<0>(val), this->matchers.at(threadid).getResult(),
                                 this->matchers.at(threadid).getInfo());
            {
                std::unique_lock<std::mutex> lockcnt(this->cnt_mutex);
                --(this->remaining_matches);
            }
            lockop.unlock();
            this->op_condition.notify_one();
        }
    }
}

template <class T, class R>
void MultiMatcher<T, R>::insert(const int &id, const PCLPointCloudPtr &src,
                                const PCLPointCloudPtr &target) {
    {
        std::unique_lock<std::mutex> lock(this->ip_mutex);
        while (this->input.size() >= static_cast<size_t>(this->queue_size)) {
            this->ip_condition.wait(lock);
        }
        this->input.emplace(id, src, target);
        {
            std::unique_lock<std::mutex> lockcnt(this->cnt_mutex);
            ++(this->remaining_matches);
        }
        lock.unlock();
        this->ip_condition.notify_one();
    }
}

template <class T, class R>
bool MultiMatcher<T, R>::done() {
    {
        std::unique_lock<std::mutex> lockcnt(this->cnt_mutex);
        if (this->remaining_matches == 0) {
            return true;
        } else {
            return false;
        }
    }
}

template <class T, class R>
bool MultiMatcher<T, R>::getResult(int *id, Eigen::Affine3d *transform, Mat6 *info) {
    std::unique_lock<std::mutex> lockop(this->op_mutex);
    if (this->output.empty()) return false;
    const auto &front = this->output.front();
    if (id) *id = std::get<0>(front);
    if (transform) *transform = std::get<1>(front);
    if (info) *info = std::get<2>(front);
    this->output.pop();
    return true;
}

}  // namespace wave

#endif  // WAVE_MULTI_MATCHER_IMPL_HPP
