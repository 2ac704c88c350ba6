// orc_throughput.cpp -- times the oracle's own cvarArMultRegistration restatement (orc_registration) frame-parallel over
// host threads: the "CPU path on the same box's host cores" that bench.py reports beside the GPU number (SURVEY 8(d):
// one thread, the way the reference runs, and all hardware threads).  TEST / MEASUREMENT INFRASTRUCTURE like the rest of
// oracle/: never part of the product path.
#include "oracle.h"
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

// Runs orc_registration (stateless: no markers carried in) on frames[i % n_frames] for i = 0, 1, ... from `threads`
// threads until `budget_s` seconds have passed (each thread finishes the frame it is on) or `max_calls` calls are done.
// Every call works on a private copy of its frame (the path greys its input in place).
// Returns the number of calls; *seconds receives the wall time from the first call's start to the last call's end.
extern "C" long long orc_registration_throughput(const uint8_t* frames, int n_frames, int w, int h, int stride, size_t frame_stride,
                                                 const OrcTemplate* templates, int n_templates, const OrcCamera* cam, int threads,
                                                 double budget_s, long long max_calls, double* seconds, int* threads_used) {
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    if (threads_used) *threads_used = threads;
    std::atomic<long long> next(0), done(0);
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    auto worker = [&]() {
        std::vector<uint8_t> img((size_t)h * stride);
        std::vector<OrcMarker> markers(512);
        for (;;) {
            const long long i = next.fetch_add(1);
            if (i >= max_calls || (i >= threads && elapsed() >= budget_s)) break;   // every thread completes at least one frame
            std::memcpy(img.data(), frames + (size_t)(i % n_frames) * frame_stride, img.size());
            orc_registration(img.data(), w, h, stride, markers.data(), 0, (int)markers.size(), templates, n_templates, cam, nullptr, 0, nullptr);
            done.fetch_add(1);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    *seconds = elapsed();
    return done.load();
}
