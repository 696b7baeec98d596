// Throughput of the drop-in wave::MultiMatcher<ICPMatcher, ICPMatcherParams> on one MI355X -- the
// reference's own way of scaling (one matcher per worker thread, wave_matching/include/wave/
// matching/multi_matcher.hpp:29-96; there every worker runs single-threaded PCL, here every worker
// feeds the GPU through its own context and stream).  Synthetic pairs (ground + walls + boxes, the
// target a shifted re-sampling), full-resolution ICP with the reference's default stopping rules.
//
//   bench_multimatcher <points per cloud> <pairs> <workers> [<workers> ...]
// prints one JSON line per worker count.  Built by libwave_amd/host/Makefile.
// Environment: BENCH_RES, BENCH_MULTISCALE = ICPMatcherParams::res / multiscale_steps (default -1 / 0);
// BENCH_QUEUE = capacity of the job queue (default 2 x workers, as small as the
// reference's default of 10 suggests; a worker takes everything that is queued -- up to 256 pairs --
// into one device launch when the clouds fit the batched path, so deep queues are what fill the GPU).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "wave/matching/gicp.hpp"
#include "wave/matching/icp.hpp"
#include "wave/matching/multi_matcher.hpp"
#include "wave/matching/ndt.hpp"

namespace {

wave::PCLPointCloudPtr scene(int n, unsigned seed, float dx) {
    auto c = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::normal_distribution<float> noise(0.f, 0.01f);
    c->points.resize((size_t) n);
    for (int i = 0; i < n; ++i) {
        pcl::PointXYZ p;
        const float pick = u(rng);
        if (pick < 0.5f) {  // ground
            p.x = -20.f + 40.f * u(rng);
            p.y = -15.f + 30.f * u(rng);
            p.z = noise(rng);
        } else if (pick < 0.8f) {  // four walls
            const int side = (int) (4.f * u(rng)) & 3;
            const float t = u(rng);
            p.x = side == 0 ? -20.f : side == 1 ? 20.f : -20.f + 40.f * t;
            p.y = side == 2 ? -15.f : side == 3 ? 15.f : -15.f + 30.f * t;
            p.z = 6.f * u(rng);
        } else {  // a few boxes
            const int b = (int) (8.f * u(rng)) & 7;
            p.x = -16.f + 4.5f * (float) b + 1.5f * u(rng);
            p.y = -10.f + 2.6f * (float) b + 1.0f * u(rng);
            p.z = 1.5f * u(rng);
        }
        p.x += dx;
        c->points[(size_t) i] = p;
    }
    return c;
}

}  // namespace

// the pool of one matcher type over the same synthetic pairs (BENCH_MATCHER = gicp | ndt: the reference's
// MultiMatcher<T, R> is generic, multi_matcher.hpp:29)
template <class M, class P>
int run_pool(const char *name, const P &params, int n, int pairs, int argc, char **argv,
             const std::vector<wave::PCLPointCloudPtr> &refs, const std::vector<wave::PCLPointCloudPtr> &targets) {
    for (int a = 3; a < argc; ++a) {
        const int workers = std::atoi(argv[a]);
        wave::MultiMatcher<M, P>::setMaxWorkers(workers);
        const char *qenv = std::getenv("BENCH_QUEUE");
        const int queue = qenv && std::atoi(qenv) > 0 ? std::atoi(qenv) : 2 * workers;
        wave::MultiMatcher<M, P> pool(workers, queue, params);
        double shift_sum = 0;
        auto drain = [&](int count) {
            int got = 0, ok = 0;
            shift_sum = 0;
            for (int j = 0; j < count; ++j) pool.insert(j, refs[(size_t) j % 4], targets[(size_t) j % 4]);
            // getResult() blocks while registrations are pending and returns false once everything inserted has
            // been handed out (the reference's documented contract, multi_matcher.hpp:64-77): no polling
            int id;
            Eigen::Affine3d T;
            wave::Mat6 info;
            while (pool.getResult(&id, &T, &info)) {
                ++got;
                ok += std::abs(T.translation()(0) - 0.15) < 0.05;
                shift_sum += T.translation()(0);
            }
            if (got != count) std::fprintf(stderr, "bench_multimatcher: %d results for %d pairs\n", got, count);
            return ok;
        };
        drain(std::max(2 * workers, std::min(queue, pairs)));
        const auto t0 = std::chrono::steady_clock::now();
        const int ok = drain(pairs);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int launches = 0;
        for (int b : pool.batchesPerSlot()) launches += b;
        std::printf("{\"bench\": \"wave::MultiMatcher<%s>\", \"points\": %d, \"workers\": %d, \"queue\": %d, \"pairs\": %d, "
                    "\"seconds\": %.4f, \"registrations_per_s\": %.1f, \"recovered_shift\": %d, \"mean_shift_x\": %.4f, "
                    "\"batches\": %d, \"pairs_per_batch\": %.1f}\n",
                    name, n, pool.workers(), queue, pairs, s, pairs / s, ok, shift_sum / pairs, launches,
                    launches ? (double) (pairs + std::max(2 * workers, std::min(queue, pairs))) / launches : 0.0);
        std::fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <points> <pairs> <workers>...\n", argv[0]);
        return 2;
    }
    const int n = std::atoi(argv[1]), pairs = std::atoi(argv[2]);
    wave::ICPMatcherParams params;
    params.res = -1;  // full resolution: the clouds go to the device as they are
    params.multiscale_steps = 0;
    // (BENCH_RES / BENCH_MULTISCALE: the reference's voxel-filtered branches, e.g. its defaults 0.1 / 3)
    if (const char *e = std::getenv("BENCH_RES")) params.res = (float) std::atof(e);
    if (const char *e = std::getenv("BENCH_MULTISCALE")) params.multiscale_steps = std::atoi(e);
    std::vector<wave::PCLPointCloudPtr> refs, targets;
    // BENCH_PAIR = resample (default: the target is an independent re-sampling of the scene, shifted by 0.15 m) or
    // copy (the target is the ref cloud shifted by 0.15 m + N(0, 1 cm) per coordinate -- what the reference's own tests
    // register, tests/icp_tests.cpp:31: with it PCL's relative-MSE stop, fit_eps 1e-2, lets sparse 10k-point pairs
    // converge, and `recovered_shift` counts them; a 10k re-sampling of a 40 x 30 m scene has its points 0.35 m apart,
    // the MSE barely moves and every pair stops half-way)
    const char *pair_env = std::getenv("BENCH_PAIR");
    const bool copy_pairs = pair_env && std::string(pair_env) == "copy";
    for (int k = 0; k < 4; ++k) {  // four distinct pairs, reused round-robin
        refs.push_back(scene(n, 100u + (unsigned) k, 0.f));
        if (copy_pairs) {
            auto t = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>(*refs.back());
            std::mt19937 rng(300u + (unsigned) k);
            std::normal_distribution<float> noise(0.f, 0.01f);
            for (auto &q : t->points) {
                q.x += 0.15f + noise(rng);
                q.y += noise(rng);
                q.z += noise(rng);
            }
            targets.push_back(t);
        } else {
            targets.push_back(scene(n, 200u + (unsigned) k, 0.15f));
        }
    }
    if (const char *e = std::getenv("BENCH_ESTIMATOR")) {  // LUM (default: LUM + Censi + LUMold run, icp.cpp:135-142) | CENSI | LUM_OLD
        const std::string v(e);
        params.covar_estimator = v == "LUM_OLD" ? wave::ICPMatcherParams::covar_method::LUMold
                                 : v == "CENSI" ? wave::ICPMatcherParams::covar_method::CENSI
                                                : wave::ICPMatcherParams::covar_method::LUM;
    }
    if (const char *m = std::getenv("BENCH_MATCHER")) {
        if (std::string(m) == "gicp") {
            wave::GICPMatcherParams gp;
            gp.res = -1;
            if (const char *e = std::getenv("BENCH_RES")) gp.res = (float) std::atof(e);
            return run_pool<wave::GICPMatcher, wave::GICPMatcherParams>("GICPMatcher", gp, n, pairs, argc, argv, refs, targets);
        }
        if (std::string(m) == "ndt") {
            wave::NDTMatcherParams np;
            np.res = 1.0f;
            if (const char *e = std::getenv("BENCH_RES")) np.res = (float) std::atof(e);
            return run_pool<wave::NDTMatcher, wave::NDTMatcherParams>("NDTMatcher", np, n, pairs, argc, argv, refs, targets);
        }
    }
    for (int a = 3; a < argc; ++a) {
        const int workers = std::atoi(argv[a]);
        wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams>::setMaxWorkers(workers);
        const char *qenv = std::getenv("BENCH_QUEUE");
        const int queue = qenv && std::atoi(qenv) > 0 ? std::atoi(qenv) : 2 * workers;
        wave::MultiMatcher<wave::ICPMatcher, wave::ICPMatcherParams> pool(workers, queue, params);
        double shift_sum = 0;
        auto drain = [&](int count) {
            int got = 0, ok = 0;
            shift_sum = 0;
            for (int j = 0; j < count; ++j) pool.insert(j, refs[(size_t) j % 4], targets[(size_t) j % 4]);
            // getResult() blocks while registrations are pending and returns false once everything inserted has
            // been handed out (the reference's documented contract, multi_matcher.hpp:64-77): no polling
            int id;
            Eigen::Affine3d T;
            wave::Mat6 info;
            while (pool.getResult(&id, &T, &info)) {
                ++got;
                ok += std::abs(T.translation()(0) - 0.15) < 0.05;
                shift_sum += T.translation()(0);
            }
            if (got != count) std::fprintf(stderr, "bench_multimatcher: %d results for %d pairs\n", got, count);
            return ok;
        };
        drain(std::max(2 * workers, std::min(queue, pairs)));  // warm-up: contexts, allocations, the tuned grid cell
        const auto t0 = std::chrono::steady_clock::now();
        const int ok = drain(pairs);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"bench\": \"wave::MultiMatcher<ICPMatcher>\", \"pair\": \"%s\", \"forced_iterations\": %d, \"res\": %g, \"multiscale_steps\": %d, \"points\": %d, \"workers\": %d, \"queue\": %d, "
                    "\"pairs\": %d, \"seconds\": %.4f, \"registrations_per_s\": %.1f, \"recovered_shift\": %d, "
                    "\"mean_shift_x\": %.4f}\n",
                    copy_pairs ? "copy + 1 cm noise" : "resample", std::getenv("WAVE_ICP_BENCH_FORCE_ITERATIONS") ? std::atoi(std::getenv("WAVE_ICP_BENCH_FORCE_ITERATIONS")) : 0,
                    (double) params.res, params.multiscale_steps, n, pool.workers(), queue, pairs, s, pairs / s, ok, shift_sum / pairs);
        std::fflush(stdout);
    }
    return 0;
}
