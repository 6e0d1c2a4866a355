// hvc_pool.h -- the host threads of an hvc_ctx (internal): a persistent pool the batch pipelines run their workers on,
// and the one place where C++ exceptions become hvc_status codes.
//
// The model is single-threaded OCaml ("no threading"); the threads here belong to the library's own
// batch entry points (hvc_jpeg_decode_batch*, hvc_jpeg_encode_batch*, the download side of the host-buffer calls).
// Round 2 created and joined `threads` std::threads in every such call, outside any try block: a caller looping over
// small batches paid for them each time, and a std::system_error from thread creation (EAGAIN under a pids limit:
// 8 ranks x 16 workers on one node) would have crossed the C boundary -- or, with some threads already started,
// ended the process in std::terminate from the vector's destructor.  Now:
//   * the workers live in the context: created on the first batch call, more added when a call asks for more,
//     joined in hvc_destroy;
//   * thread creation failing is HVC_E_SYSTEM, the threads that did start stay usable;
//   * a task that throws is caught on its thread and reported by wait();
//   * every extern "C" body is a function-try-block ending in HVC_ABI_CATCH.
#ifndef HVC_POOL_H
#define HVC_POOL_H

#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/hvc_jpeg.h"

namespace hvc {

// the current exception as a status code (call inside a catch block only)
inline int exception_code() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    } catch (const std::system_error &) {
        return HVC_E_SYSTEM;
    } catch (...) {
        return HVC_E_INTERNAL;
    }
}

class WorkerPool {
  public:
    WorkerPool() = default;
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    ~WorkerPool() { shutdown(); }

    int size() const noexcept { return (int)threads_.size(); }
    unsigned long long threads_created() const noexcept { return created_; }

    // At least n threads.  HVC_OK, HVC_E_SYSTEM (the system refused a thread) or HVC_E_OUT_OF_MEMORY; the threads that
    // exist stay.  HVC_POOL_FAIL_AFTER=k (tests): the k-th thread creation of the process and all later ones fail the
    // way a pids limit makes them fail.
    int ensure(int n) noexcept {
        try {
            threads_.reserve((size_t)n);
            while ((int)threads_.size() < n) {
                if (fail_after() >= 0 && (long long)global_created()++ >= fail_after())
                    throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
                threads_.emplace_back([this] { run(); });
                created_++;
            }
            return HVC_OK;
        } catch (...) {
            return exception_code();
        }
    }

    // `copies` pool threads will each call fn once, concurrently if the pool has that many idle threads (ensure() first).
    // Returns at once; wait() before anything fn refers to goes out of scope.
    int submit(const std::function<void()> &fn, int copies) noexcept {
        try {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < copies; i++) tasks_.push_back(fn);
            pending_ += copies;
            cv_task_.notify_all();
            return HVC_OK;
        } catch (...) {
            return exception_code();
        }
    }

    // All submitted tasks have returned.  HVC_OK, or the code of an exception one of them ended with.
    int wait() noexcept {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this] { return pending_ == 0; });
        const int e = task_error_;
        task_error_ = HVC_OK;
        return e;
    }

    void shutdown() noexcept {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            cv_task_.notify_all();
        }
        for (std::thread &t : threads_) {
            try {
                if (t.joinable()) t.join();
            } catch (...) {
            }
        }
        threads_.clear();
        stop_ = false;
    }

  private:
    static long long fail_after() noexcept {
        static const long long v = [] {
            const char *e = std::getenv("HVC_POOL_FAIL_AFTER");
            return e ? std::atoll(e) : -1ll;
        }();
        return v;
    }
    static unsigned long long &global_created() noexcept {
        static unsigned long long n = 0;
        return n;
    }
    void run() noexcept {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_task_.wait(lk, [this] { return stop_ || !tasks_.empty(); });
                if (tasks_.empty()) return; // stop
                fn = std::move(tasks_.front());
                tasks_.pop_front();
            }
            int e = HVC_OK;
            try {
                fn();
            } catch (...) {
                e = exception_code();
            }
            std::lock_guard<std::mutex> lk(mu_);
            if (e && !task_error_) task_error_ = e;
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }

    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> tasks_;
    std::mutex mu_;
    std::condition_variable cv_task_, cv_done_;
    int pending_ = 0, task_error_ = HVC_OK;
    bool stop_ = false;
    unsigned long long created_ = 0;
};

// Waits for the pool's tasks when the scope ends, however it ends: the tasks refer to the caller's locals.  `on_exit`
// (optional) runs first -- the place to raise the pipeline's error flag and wake the workers so that they do end.
class PoolScope {
  public:
    PoolScope(WorkerPool &p, std::function<void()> on_exit) : pool_(p), on_exit_(std::move(on_exit)) {}
    PoolScope(const PoolScope &) = delete;
    PoolScope &operator=(const PoolScope &) = delete;
    ~PoolScope() { (void)finish(); }
    int finish() noexcept { // idempotent; returns the tasks' exception code, if any
        if (done_) return code_;
        done_ = true;
        try {
            if (on_exit_) on_exit_();
        } catch (...) {
        }
        code_ = pool_.wait();
        return code_;
    }

  private:
    WorkerPool &pool_;
    std::function<void()> on_exit_;
    bool done_ = false;
    int code_ = HVC_OK;
};

} // namespace hvc

// every extern "C" function is a function-try-block:  int hvc_f(...) try { ... } HVC_ABI_CATCH
#define HVC_ABI_CATCH \
    catch (...) { return hvc::exception_code(); }

#endif
