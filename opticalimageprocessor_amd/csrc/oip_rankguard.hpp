// oip_rankguard.hpp -- how the rank threads of the N-GPU host (oip_multigpu.hpp) meet, fail and give up together.
// No GPU or RCCL types in here: tests/test_cli_cpu.py builds it with ThreadSanitizer on the CPU.
//
//   HostBarrier   the ranks' meeting point; abort() releases everybody (wait() returns false from then on).
//   CommGuard     the communicators.  Every use of a communicator -- a single call or a whole ncclGroupStart..End
//                 sequence -- runs inside use(), which holds the guard shared; abort_all() takes it exclusively, so it
//                 waits for the calls in flight, runs the abort function on every communicator exactly once, and every
//                 later use() throws PeerFailed instead of touching a communicator that ncclCommAbort has freed
//                 (the failing thread used to abort while its peers were still issuing ncclSend / ncclRecv).
#pragma once

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <vector>

namespace OIPGPU {

struct PeerFailed : public std::runtime_error {
    PeerFailed() : std::runtime_error("another GPU's step failed") {}
};

struct HostBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long gen = 0;
    bool aborted = false;
    explicit HostBarrier(int n_) : n(n_) {}
    bool wait()                 // false: a rank has given up (abort()), nobody waits any longer
    {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort()
    {
        std::unique_lock<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

template <typename Comm> class CommGuard {
public:
    explicit CommGuard(int n) : comm((size_t)n, Comm()) {}
    std::vector<Comm> comm;                     // filled by the owner before the rank threads start
    // f(communicator of rank r) under the shared lock; PeerFailed once the communicators are gone
    template <typename F> void use(int r, F f)
    {
        std::shared_lock<std::shared_mutex> lk(mMu);
        if (mAborted.load()) throw PeerFailed();
        f(comm[(size_t)r]);
    }
    // abort_one(communicator) for every communicator, once, after the calls in flight have returned
    template <typename A> void abort_all(A abort_one)
    {
        std::unique_lock<std::shared_mutex> lk(mMu);
        if (mAborted.exchange(true)) return;
        for (auto &c : comm) abort_one(c);
    }
    bool aborted() const { return mAborted.load(); }

private:
    std::shared_mutex mMu;
    std::atomic<bool> mAborted{false};
};

}  // namespace OIPGPU
