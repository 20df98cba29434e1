// encode_host.cpp - the input encoder of the reasoners as native host code: per-frame detections -> boxes [T][15][F] fp32
// + the heuristic "object to track" index vector [T].  Host side of libopnet_hip.so (no GPU work, no HIP call).
//
// What is computed (reference baselines/datasets.py):
//   _normalize_and_pad_predictions   :130-196 (5 tracks), :265-336 (6 tracks)
//   _get_closest_object_to_track_vector :199-257 (5 tracks), :338-416 (6 tracks)
// exactly as objectpermanence_amd/datasets.py restates them (encode_boxes / index_to_track - the numpy form stays the
// readable statement and the fallback; this file is what the dataset classes run):
//   slot order   = the clip's distinct class ids, snitch (140) first, then ascending (:47-54, :271-274), 15 slots (:291-292);
//   slot content = the FIRST detection of that id in the frame (:294-309) - for the snitch the LAST one (the reference's
//                  comparator returns -1 for (snitch, snitch) either way, so its insertion sort reverses repeated snitches);
//                  [x1/320, y1/240, x2/320, y2/240, 1 (, is_cone)] - float64 division, then the cast to fp32 the reference's
//                  torch.tensor(dtype=float32) performs; a missing object is zeros, except that a missing CONE keeps its cone
//                  bit while the reference's walk still has detections to place (slot < the frame's largest rank, :311-318);
//   index vector = the containment-stack state machine on the float64 boxes, centres and distances in double, argmin = first
//                  minimum (numpy.argmin), no fused multiply-add (numpy evaluates d0*d0 + d1*d1 in two roundings).
// Bit-exact against tests/golden/datasets.npz and the 300-case fuzz against the reference's own classes
// (tests/test_datasets.py, oracle/fuzz_datasets.py).  The reference takes 16.7 ms per clip, the numpy form 1.2 ms, this ~15 us.
// (built twice: into libopnet_hip.so with hipcc - the C ABI's entry point - and alone into libopnet_encode.so with g++
// -ffp-contract=off, which DataLoader worker processes load without paying for the HIP runtime's start-up)
#ifdef __clang__
#pragma clang fp contract(off)
#endif

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define ENC_SLOTS 15
#define ENC_SNITCH 140
#define ENC_LUT 4096            // class ids below this take the table path

extern "C" __attribute__((visibility("default")))
int opnet_encode_clips_f32(const int32_t *counts, const int32_t *ids, const int32_t *bb, const int64_t *clip_first_frame,
                           const int64_t *clip_first_det, int n_clips, int T, int n_tracks, const uint8_t *is_cone, int n_classes,
                           float *boxes_out, int64_t *index_out)
{
    // counts [sum T]: detections per frame; ids [N]; bb [N][4] (x1, y1, x2, y2 pixels); clip c owns frames
    // clip_first_frame[c] .. +T and detections clip_first_det[c] .. clip_first_det[c + 1]; is_cone [n_classes] (ids beyond
    // the table are not cones); boxes_out [n_clips][T][15][n_tracks]; index_out [n_clips][T] (may be null)
    if (!counts || !ids || !bb || !clip_first_frame || !clip_first_det || !is_cone || !boxes_out) return -1;
    if (n_clips < 0 || T <= 0 || (n_tracks != 5 && n_tracks != 6)) return -3;
    const int F = n_tracks;
    std::vector<int32_t> uniq;
    static thread_local int16_t lut[ENC_LUT];
    std::vector<double> box64((size_t)T * ENC_SLOTS * 6);
    for (int c = 0; c < n_clips; ++c) {
        const int32_t *cnt = counts + clip_first_frame[c];
        const int64_t d0 = clip_first_det[c], d1 = clip_first_det[c + 1];
        const int64_t n = d1 - d0;
        const int32_t *cid = ids + d0;
        const int32_t *cbb = bb + d0 * 4;
        {
            int64_t s = 0;
            for (int t = 0; t < T; ++t) { if (cnt[t] < 0) return -3; s += cnt[t]; }
            if (s != n) return -3;
        }
        // ---- slot order ----------------------------------------------------------------------------------------------
        // class ids are small non-negative integers (193 classes): a presence table gives the sorted distinct ids and an O(1)
        // rank lookup; anything else (a negative or huge id) takes the sort + binary search path
        int32_t lo = 0, hi = -1;
        for (int64_t q = 0; q < n; ++q) { if (q == 0 || cid[q] < lo) lo = cid[q]; if (q == 0 || cid[q] > hi) hi = cid[q]; }
        const bool small = n > 0 && lo >= 0 && hi < ENC_LUT;
        uniq.clear();
        if (small) {
            std::fill(lut, lut + hi + 1, (int16_t)-1);
            for (int64_t q = 0; q < n; ++q) lut[cid[q]] = 0;
            if (hi >= ENC_SNITCH && lut[ENC_SNITCH] == 0) uniq.push_back((int32_t)ENC_SNITCH);
            for (int32_t v = 0; v <= hi; ++v)
                if (lut[v] == 0 && v != ENC_SNITCH) uniq.push_back(v);
            for (size_t r = 0; r < uniq.size(); ++r) lut[uniq[r]] = (int16_t)(r < 32767 ? r : 32767);
        } else {
            uniq.assign(cid, cid + n);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            const auto sn = std::lower_bound(uniq.begin(), uniq.end(), (int32_t)ENC_SNITCH);
            if (sn != uniq.end() && *sn == ENC_SNITCH) { uniq.erase(sn); uniq.insert(uniq.begin(), (int32_t)ENC_SNITCH); }
        }
        const bool has_snitch = !uniq.empty() && uniq[0] == ENC_SNITCH;
        const int nu = (int)uniq.size();
        const int tail0 = has_snitch ? 1 : 0;
        auto rank = [&](int32_t id) -> int {
            if (small) return lut[id];
            if (has_snitch && id == ENC_SNITCH) return 0;
            return (int)(std::lower_bound(uniq.begin() + tail0, uniq.end(), id) - uniq.begin());
        };
        const int nslot = nu < ENC_SLOTS ? nu : ENC_SLOTS;
        double cone[ENC_SLOTS];
        for (int s = 0; s < ENC_SLOTS; ++s)
            cone[s] = (s < nslot && uniq[s] >= 0 && uniq[s] < n_classes && is_cone[uniq[s]]) ? 1.0 : 0.0;
        // ---- boxes (float64, as the reference's numpy arrays) ------------------------------------------------------------
        std::fill(box64.begin(), box64.end(), 0.0);
        int64_t k = 0;
        for (int t = 0; t < T; ++t) {
            double *fr = &box64[(size_t)t * ENC_SLOTS * 6];
            bool filled[ENC_SLOTS] = {false};
            int last_rank = -1;
            for (int q = 0; q < cnt[t]; ++q, ++k) {
                const int r = rank(cid[k]);
                if (r > last_rank) last_rank = r;
                if (r >= ENC_SLOTS) continue;
                if (filled[r] && r != 0) continue;                 // first occurrence ...
                if (filled[r] && !(has_snitch && r == 0)) continue;   // ... except a repeated snitch: the last one
                filled[r] = true;
                double *o = fr + r * 6;
                o[0] = (double)cbb[k * 4 + 0] / 320.0;
                o[1] = (double)cbb[k * 4 + 1] / 240.0;
                o[2] = (double)cbb[k * 4 + 2] / 320.0;
                o[3] = (double)cbb[k * 4 + 3] / 240.0;
                o[4] = 1.0;
                o[5] = cone[r];
            }
            if (F == 6)
                for (int s = 0; s < nslot; ++s)
                    if (!filled[s] && s < last_rank) fr[s * 6 + 5] = cone[s];
        }
        float *out = boxes_out + (size_t)c * T * ENC_SLOTS * F;
        for (size_t i = 0; i < (size_t)T * ENC_SLOTS; ++i)
            for (int f = 0; f < F; ++f) out[i * F + f] = (float)box64[i * 6 + f];
        if (!index_out) continue;
        // ---- index vector: the containment stack (datasets.py:199-257 / :338-416) -------------------------------------------
        int64_t *idx = index_out + (size_t)c * T;
        const bool six = F == 6;
        std::vector<int> stack;
        double last[4] = {0.0, 0.0, 0.0, 0.0};
        int cur = 0;
        auto closest = [&](const double *fr) -> int {
            const double lx = (last[0] + last[2]) / 2, ly = (last[1] + last[3]) / 2;
            int best = 0;
            double bv = 0.0;
            for (int s = 0; s < ENC_SLOTS; ++s) {
                const double cx = (fr[s * 6 + 0] + fr[s * 6 + 2]) / 2, cy = (fr[s * 6 + 1] + fr[s * 6 + 3]) / 2;
                const double e0 = cx - lx, e1 = cy - ly;
                const double v = sqrt(e0 * e0 + e1 * e1);
                if (s == 0 || v < bv) { bv = v; best = s; }
            }
            return best;
        };
        auto take = [&](const double *row) { last[0] = row[0]; last[1] = row[1]; last[2] = row[2]; last[3] = row[3]; };
        for (int t = 0; t < T; ++t) {
            const double *fr = &box64[(size_t)t * ENC_SLOTS * 6];
            if (fr[4] != 0.0) {
                idx[t] = 0; take(fr); cur = 0; stack.clear();
            } else if (cur == 0) {
                const int cc = closest(fr);
                if (six && fr[cc * 6 + 5] == 0.0) idx[t] = 0;          // occlusion by a non-cone: keep the snitch
                else { idx[t] = cc; take(fr + cc * 6); cur = cc; stack.push_back(0); }
            } else if (fr[cur * 6 + 4] == 0.0) {
                const int cc = closest(fr);
                if (six && fr[cc * 6 + 5] == 0.0) idx[t] = cur;
                else { idx[t] = cc; take(fr + cc * 6); stack.push_back(cur); cur = cc; }
            } else {
                const int prev = stack.back();
                if (fr[prev * 6 + 4] != 0.0) { stack.pop_back(); idx[t] = prev; take(fr + prev * 6); cur = prev; }
                else { idx[t] = cur; take(fr + cur * 6); }
            }
        }
    }
    return 0;
}
