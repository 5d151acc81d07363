// copy_pool.cpp — see copy_pool.h
#include "copy_pool.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace mcrt {
namespace {

// created on first use, joined when the library is unloaded
class CopyPool {
public:
    static CopyPool& get() {
        static CopyPool pool;
        return pool;
    }
    // dst[0 .. bytes) = src[0 .. bytes), split over the workers and the calling thread; returns when done
    void copy(void* dst, const void* src, size_t bytes) {
        const size_t grain = static_cast<size_t>(512) << 10;
        const int parts = static_cast<int>(std::min<size_t>(static_cast<size_t>(workers_.size()) + 1, (bytes + grain - 1) / grain));
        if (parts <= 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::unique_lock<std::mutex> lock(mu_);
        cv_idle_.wait(lock, [&] { return !busy_; });  // one job at a time: callers from several threads take turns
        busy_ = true;  // (a new job may only start once this caller has SEEN its own job complete)
        dst_ = static_cast<char*>(dst), src_ = static_cast<const char*>(src), bytes_ = bytes, parts_ = parts;
        next_ = 1;  // part 0 is the caller's
        pending_ = parts;
        ++generation_;
        lock.unlock();
        cv_work_.notify_all();
        run_part(0);
        lock.lock();
        if (--pending_ != 0) cv_done_.wait(lock, [&] { return pending_ == 0; });
        busy_ = false;
        lock.unlock();
        cv_idle_.notify_one();
    }

private:
    CopyPool() {
        static const int n = [] {
            const char* e = std::getenv("MCRT_COPY_THREADS");
            int v = e ? std::atoi(e) : 8;
            const int hw = static_cast<int>(std::thread::hardware_concurrency());
            if (hw > 0 && v > hw) v = hw;
            return v < 1 ? 1 : v;
        }();
        for (int i = 0; i < n - 1; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    void run_part(int i) {
        const size_t per = ((bytes_ + static_cast<size_t>(parts_) - 1) / static_cast<size_t>(parts_) + 63) & ~static_cast<size_t>(63);
        const size_t a = std::min(bytes_, per * static_cast<size_t>(i)), b = std::min(bytes_, a + per);
        if (b > a) std::memcpy(dst_ + a, src_ + a, b - a);
    }
    void loop() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            cv_work_.wait(lock, [&] { return stop_ || (generation_ != seen && next_ < parts_); });
            if (stop_) return;
            const unsigned long long gen = generation_;
            while (generation_ == gen && next_ < parts_) {
                const int i = next_++;
                lock.unlock();
                run_part(i);
                lock.lock();
                if (--pending_ == 0) cv_done_.notify_all();
            }
            seen = gen;
        }
    }
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_, cv_idle_;
    std::vector<std::thread> workers_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0;
    int parts_ = 0, next_ = 0, pending_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false, busy_ = false;
};


}  // namespace

void parallel_copy(void* dst, const void* src, size_t bytes) { CopyPool::get().copy(dst, src, bytes); }

}  // namespace mcrt
