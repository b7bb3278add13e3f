// The C-ABI sharded matcher (orbx_sharded_*) with a real RCCL communicator: `world` ranks as threads of
// this process, rank r on GPU r, each holding a contiguous shard of the database.  Every rank's result must
// equal (a) orbx_match_top2_device over the whole database on one GPU and (b) a host loop with the
// bookkeeping of src/ORBmatcher.cc:217-242 (strict '<', first row wins).  A one-GPU box runs world = 1.
//   test_sharded_rccl [world] [queries] [database rows]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "orbx.hpp"

#define HIPOK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

int main(int argc, char **argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    const int nq = argc > 2 ? atoi(argv[2]) : 777;
    const int64_t ndb = argc > 3 ? atoll(argv[3]) : 50000;
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    if (world < 1 || world > ndev) {
        fprintf(stderr, "world %d needs that many GPUs (%d visible): RCCL takes one rank per device\n", world, ndev);
        return 2;
    }
    std::vector<uint8_t> db((size_t)ndb * 32), q((size_t)nq * 32);
    orbx_synth_descriptors(11, ndb, db.data());
    // queries: database rows with a few bytes changed; every fifth query has an exact copy of its source row
    // planted in the last shard, so that ties between shards are decided (the lower row must win)
    for (int i = 0; i < nq; i++) {
        const int64_t src = ((int64_t)i * 7919) % (ndb / 2);
        memcpy(&q[(size_t)i * 32], &db[(size_t)src * 32], 32);
        q[(size_t)i * 32 + (i % 32)] ^= (uint8_t)(i % 7);
        if (i % 5 == 0)
            memcpy(&db[(size_t)(ndb - 1 - i) * 32], &db[(size_t)src * 32], 32);
    }
    // (b) host loop
    std::vector<int32_t> want_idx(nq);
    std::vector<uint16_t> want_d1(nq), want_d2(nq);
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int64_t r = 0; r < ndb; r++) {
            const int d = orbx_descriptor_distance(&q[(size_t)i * 32], &db[(size_t)r * 32]);
            if (d < b1)
                b2 = b1, b1 = d, bi = (int)r;
            else if (d < b2)
                b2 = d;
        }
        want_idx[i] = bi, want_d1[i] = (uint16_t)b1, want_d2[i] = (uint16_t)b2;
    }
    // (a) one GPU, whole database
    {
        HIPOK(hipSetDevice(0));
        uint8_t *d_db, *d_q;
        int32_t *d_i;
        uint16_t *d_1, *d_2;
        HIPOK(hipMalloc((void **)&d_db, db.size()));
        HIPOK(hipMalloc((void **)&d_q, q.size()));
        HIPOK(hipMalloc((void **)&d_i, nq * 4));
        HIPOK(hipMalloc((void **)&d_1, nq * 2));
        HIPOK(hipMalloc((void **)&d_2, nq * 2));
        HIPOK(hipMemcpy(d_db, db.data(), db.size(), hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(d_q, q.data(), q.size(), hipMemcpyHostToDevice));
        orbx::Matcher m(0.75f, true, 0, nq);
        orbx::check(orbx_match_top2_device(m.handle(), d_q, nq, d_db, ndb, 0, d_i, d_1, d_2), "orbx_match_top2_device");
        HIPOK(hipDeviceSynchronize());
        std::vector<int32_t> gi(nq);
        std::vector<uint16_t> g1(nq), g2(nq);
        HIPOK(hipMemcpy(gi.data(), d_i, nq * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(g1.data(), d_1, nq * 2, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(g2.data(), d_2, nq * 2, hipMemcpyDeviceToHost));
        if (gi != want_idx || g1 != want_d1 || g2 != want_d2) {
            printf("FAIL single-GPU pass differs from the host loop\n");
            return 1;
        }
        for (void *p : {(void *)d_db, (void *)d_q, (void *)d_i, (void *)d_1, (void *)d_2})
            HIPOK(hipFree(p));
    }
    // the sharded call, twice per rank (the second with fewer queries: buffers are reused)
    orbx::ShardedMatcher::UniqueId id;
    try {
        id = orbx::ShardedMatcher::unique_id();
    } catch (const orbx::Error &e) {
        printf("FAIL %s\n", e.what());
        return 1;
    }
    std::vector<std::string> fail(world);
    std::vector<std::thread> th;
    const int64_t per = (ndb + world - 1) / world;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r] {
            try {
                HIPOK(hipSetDevice(r));
                const int64_t lo = std::min<int64_t>(r * per, ndb), hi = std::min<int64_t>(lo + per, ndb);
                uint8_t *d_db, *d_q;
                int32_t *d_i;
                uint16_t *d_1, *d_2;
                HIPOK(hipMalloc((void **)&d_db, std::max<size_t>((size_t)(hi - lo) * 32, 32)));
                HIPOK(hipMalloc((void **)&d_q, q.size()));
                HIPOK(hipMalloc((void **)&d_i, nq * 4));
                HIPOK(hipMalloc((void **)&d_1, nq * 2));
                HIPOK(hipMalloc((void **)&d_2, nq * 2));
                HIPOK(hipMemcpy(d_db, &db[(size_t)lo * 32], (size_t)(hi - lo) * 32, hipMemcpyHostToDevice));
                HIPOK(hipMemcpy(d_q, q.data(), q.size(), hipMemcpyHostToDevice));
                orbx::ShardedMatcher sm(r, r, world, id, nq);
                for (int pass = 0; pass < 2; pass++) {
                    const int n = pass ? nq / 3 + 1 : nq;
                    sm.match_device(d_q, n, d_db, hi - lo, lo, d_i, d_1, d_2);
                    sm.synchronize();
                    std::vector<int32_t> gi(n);
                    std::vector<uint16_t> g1(n), g2(n);
                    HIPOK(hipMemcpy(gi.data(), d_i, n * 4, hipMemcpyDeviceToHost));
                    HIPOK(hipMemcpy(g1.data(), d_1, n * 2, hipMemcpyDeviceToHost));
                    HIPOK(hipMemcpy(g2.data(), d_2, n * 2, hipMemcpyDeviceToHost));
                    for (int i = 0; i < n; i++)
                        if (gi[i] != want_idx[i] || g1[i] != want_d1[i] || g2[i] != want_d2[i]) {
                            char buf[160];
                            snprintf(buf, sizeof(buf), "rank %d pass %d query %d: got (%d,%d,%d) want (%d,%d,%d)", r, pass,
                                     i, gi[i], g1[i], g2[i], want_idx[i], want_d1[i], want_d2[i]);
                            fail[r] = buf;
                            break;
                        }
                }
                for (void *p : {(void *)d_db, (void *)d_q, (void *)d_i, (void *)d_1, (void *)d_2})
            HIPOK(hipFree(p));
            } catch (const orbx::Error &e) {
                fail[r] = e.what();
            }
        });
    for (auto &t : th)
        t.join();
    for (int r = 0; r < world; r++)
        if (!fail[r].empty()) {
            printf("FAIL %s\n", fail[r].c_str());
            return 1;
        }
    int planted = 0;
    for (int i = 0; i < nq; i += 5)
        planted += want_d1[i] <= 6 && want_idx[i] < ndb / 2;
    printf("OK world=%d nq=%d ndb=%lld ties_to_lower_row=%d\n", world, nq, (long long)ndb, planted);
    return 0;
}
