// rankguard_test.cpp -- the failure protocol of the N-GPU host (csrc/oip_rankguard.hpp) on the CPU, under ThreadSanitizer:
// N rank threads pass the pre-exchange barrier and keep posting "RCCL calls" on fake communicators; one rank fails right
// behind the barrier and aborts them all (a peer used to be able to touch a communicator that ncclCommAbort had
// freed).  A fake communicator is a heap object: "abort" deletes it, a "call" reads and writes it -- so a use after the
// abort is a use after free that TSan / ASan report, and the test checks the counters as well.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "oip_rankguard.hpp"

using namespace OIPGPU;

struct FakeComm {
    std::atomic<long> calls{0};
    bool alive = true;
};

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 200;
    int bad = 0;
    for (int round = 0; round < rounds; ++round) {
        HostBarrier bar(N);
        CommGuard<FakeComm *> comms(N);
        for (auto &c : comms.comm) c = new FakeComm();
        std::atomic<bool> failed{false};
        std::atomic<int> peer_failed{0}, own_failure{0}, used_after_abort{0};
        const int culprit = round % N;
        std::vector<std::thread> th;
        for (int r = 0; r < N; ++r)
            th.emplace_back([&, r] {
                try {
                    if (!bar.wait()) throw PeerFailed();                  // the barrier that precedes the exchange
                    if (r == culprit) {
                        std::this_thread::sleep_for(std::chrono::microseconds(50 * (round % 7)));
                        throw std::runtime_error("injected");
                    }
                    for (int g = 0; g < 2000; ++g) {                      // groups of posted transfers
                        if (failed) throw PeerFailed();
                        comms.use(r, [&](FakeComm *c) {
                            if (!c->alive) ++used_after_abort;            // (would also be a heap-use-after-free)
                            for (int k = 0; k < 8; ++k) ++c->calls;       // ncclGroupStart, sends, recvs, ncclGroupEnd
                        });
                    }
                    if (!bar.wait()) throw PeerFailed();                  // finish_pieces' meeting point
                } catch (const PeerFailed &) {
                    ++peer_failed;
                    failed = true; bar.abort();
                    comms.abort_all([](FakeComm *c) { c->alive = false; delete c; });
                } catch (const std::exception &) {
                    ++own_failure;
                    failed = true; bar.abort();
                    comms.abort_all([](FakeComm *c) { c->alive = false; delete c; });
                }
            });
        for (auto &t : th) t.join();
        if (own_failure != 1 || peer_failed != N - 1 || used_after_abort != 0 || !comms.aborted()) {
            printf("round %d: own %d peers %d used-after-abort %d\n", round, own_failure.load(), peer_failed.load(), used_after_abort.load());
            ++bad;
        }
    }
    printf("%d rounds, %d bad\n", rounds, bad);
    return bad ? 1 : 0;
}
