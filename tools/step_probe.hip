// step_probe.hip - in-kernel timeline of opnet_step (built with -DOPNET_TRACE, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOPNET_TRACE -o tools/step_probe tools/step_probe.hip
#include "../objectpermanence_amd/csrc/opnet_abi.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32, T = 300, H1 = 256, H2 = 512;
    const size_t nb = (size_t)B * T * 90;
    std::vector<float> hb(nb);
    for (size_t i = 0; i < nb; ++i) hb[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
    auto randw = [](size_t n, float bound) { std::vector<float> v(n); for (size_t i = 0; i < n; ++i) v[i] = ((float)((i * 40503u + 17) % 2001) / 1000.f - 1.f) * bound; return v; };
    std::vector<float> wih1 = randw(4 * H1 * 90, .1f), whh1 = randw(4 * H1 * H1, .1f), wsel = randw(15 * H1, .5f),
                       wih2 = randw(4 * H2 * 6, 1.f), whh2 = randw(4 * H2 * H2, .08f), wout = randw(4 * H2, .3f);
    float *d_b, *d_w[6], *d_packed, *d_y, *d_lg; void *d_ws;
    CK(hipMalloc(&d_b, nb * 4)); CK(hipMemcpy(d_b, hb.data(), nb * 4, hipMemcpyHostToDevice));
    std::vector<float> *ws[6] = {&wih1, &whh1, &wsel, &wih2, &whh2, &wout};
    for (int i = 0; i < 6; ++i) { CK(hipMalloc(&d_w[i], ws[i]->size() * 4)); CK(hipMemcpy(d_w[i], ws[i]->data(), ws[i]->size() * 4, hipMemcpyHostToDevice)); }
    const size_t pb = opnet_packed_weights_bytes(H1, H2), wb = opnet_workspace_bytes(B, T, H1, H2);
    CK(hipMalloc(&d_packed, pb)); CK(hipMalloc(&d_ws, wb)); CK(hipMalloc(&d_y, (size_t)B * T * 16)); CK(hipMalloc(&d_lg, (size_t)B * T * 60));
    if (opnet_pack_weights_f32(d_w[0], d_w[1], d_w[2], d_w[3], d_w[4], d_w[5], d_packed, pb, H1, H2, nullptr)) { printf("%s\n", opnet_last_error()); return 1; }
    const int RB = (B + 31) / 32, NWG = (H2 / 4 + H1 / 4 + 2);
    const size_t ntrace = (size_t)(T + 3) * NWG * RB * 8;
    unsigned long long *d_trace; CK(hipMalloc(&d_trace, ntrace * 8)); CK(hipMemset(d_trace, 0, ntrace * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &d_trace, sizeof(d_trace)));
    opnet_plan *plan; opnet_plan_create(&plan, B, T, H1, H2);
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int it = 0; it < 5; ++it) {
        if (opnet_plan_forward(plan, d_b, d_packed, d_y, d_lg, d_ws, wb, st)) { printf("%s\n", opnet_last_error()); return 1; }
    }
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < 10; ++it) opnet_plan_forward(plan, d_b, d_packed, d_y, d_lg, d_ws, wb, st);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d: %.3f ms/forward (traced build), %.2f us/step\n", B, ms / 10, ms / 10 / (T + 3) * 1e3);
    std::vector<unsigned long long> tr(ntrace);
    CK(hipMemcpy(tr.data(), d_trace, ntrace * 8, hipMemcpyDeviceToHost));
    // per-role statistics over steps 10..290, row block 0
    const char *names[4] = {"LSTM2 tile", "LSTM1 tile", "sel head", "out head"};
    int first[4] = {0, H2 / 4, H2 / 4 + H1 / 4, H2 / 4 + H1 / 4 + 1};
    int count[4] = {H2 / 4, H1 / 4, 1, 1};
    for (int r = 0; r < 4; ++r) {
        double seg[5] = {0, 0, 0, 0, 0}; long n = 0;
        for (int s = 10; s < 290; ++s) for (int w = first[r]; w < first[r] + count[r]; ++w) {
            const unsigned long long *p = &tr[(((size_t)s * NWG + w) * RB + 0) * 8];
            if (!p[5]) continue;
            seg[0] += (double)(p[2] - p[1]); seg[1] += (double)(p[3] - p[2]); seg[2] += (double)(p[4] - p[3]); seg[3] += (double)(p[5] - p[4]); seg[4] += (double)(p[5] - p[1]);
            ++n;
        }
        printf("%-11s cycles(s_memtime): entry->loads landed %.0f | mfma+lds write %.0f | barrier %.0f | epilogue %.0f | total %.0f  (n=%ld)\n",
               names[r], seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, seg[4] / n, n);
    }
    // launch skew and step period from the 100 MHz wall clock
    double period = 0, skew = 0; int np = 0;
    for (int s = 10; s < 290; ++s) {
        unsigned long long mn = ~0ull, mx = 0;
        for (int w = 0; w < NWG; ++w) { unsigned long long v = tr[(((size_t)s * NWG + w) * RB) * 8]; if (v) { mn = std::min(mn, v); mx = std::max(mx, v); } }
        unsigned long long mn2 = ~0ull;
        for (int w = 0; w < NWG; ++w) { unsigned long long v = tr[(((size_t)(s + 1) * NWG + w) * RB) * 8]; if (v) mn2 = std::min(mn2, v); }
        period += (double)(mn2 - mn); skew += (double)(mx - mn); ++np;
    }
    printf("wall clock (100 MHz ticks): step period %.1f ticks = %.2f us, first->last workgroup start skew %.1f ticks\n", period / np, period / np / 100.0, skew / np);
    return 0;
}
