// Test-only tool: independent check of the XORWOW recurrence and the 2^67 sequence
// skip against rocRAND's engine (third-party, /opt/rocm/include/rocrand).  Reads
// lines "d v0 v1 v2 v3 v4 subsequence nsteps" and prints the state after
// discard_subsequence(subsequence) followed by nsteps next() calls, plus the last output.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <cstdio>

struct Engine : public rocrand_device::xorwow_engine {
    __host__ void set(const unsigned v[5], unsigned d) { for (int i = 0; i < 5; i++) m_state.x[i] = v[i]; m_state.d = d; }
    __host__ void get(unsigned v[5], unsigned &d) const { for (int i = 0; i < 5; i++) v[i] = m_state.x[i]; d = m_state.d; }
};

int main() {
    unsigned d, v[5];
    unsigned long long sub;
    int nsteps;
    while (std::scanf("%u %u %u %u %u %u %llu %d", &d, &v[0], &v[1], &v[2], &v[3], &v[4], &sub, &nsteps) == 8) {
        Engine e;
        e.set(v, d);
        e.discard_subsequence(sub);
        unsigned last = 0;
        for (int i = 0; i < nsteps; i++) last = e.next();
        e.get(v, d);
        std::printf("%u %u %u %u %u %u %u\n", d, v[0], v[1], v[2], v[3], v[4], last);
    }
    return 0;
}
