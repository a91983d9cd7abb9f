// savad_post.h -- host-side post-processing of the reference's predict path (scope table next-row 3):
// frame smoothing, frame -> sample overlap-average, sample -> segments, optimal split of long
// activities.  O(N) scalar state machines on host arrays (they run after the device -> host copy of
// the per-frame probabilities), restated with their quirks; pinned against goldens produced by the
// reference functions themselves (tests/golden/make_golden_post.py).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace savad {
namespace post {

// vad/postprocessing/trim.py:4-66.  pred / out: n values in {0,1}.
inline void trim_voice_activity(const uint8_t* pred, int n, int min_vally, int min_hill, int hang_before, int hang_over,
                                uint8_t* out) {
    std::copy(pred, pred + n, out);
    std::vector<uint8_t> snap;
    auto fill = [&](long lo, long hi, uint8_t v) {  // python slice assignment out[lo:hi] = v (clipped, negatives wrap)
        if (lo < 0) lo = std::max<long>(0, lo + n);
        if (hi < 0) hi = std::max<long>(0, hi + n);
        lo = std::min<long>(lo, n);
        hi = std::min<long>(hi, n);
        for (long i = lo; i < hi; ++i) out[i] = v;
    };
    // every pass iterates over a SNAPSHOT (predictions_copy.tolist()) while editing the array
    if (min_vally > 0) {  // fill valleys shorter than min_vally (:15-30)
        snap.assign(out, out + n);
        bool offset = false;
        long offset_point = 0;
        for (int idx = 1; idx < n; ++idx) {  // (current, next) = (snap[idx-1], snap[idx]); idx 0 has current = None
            const uint8_t cur = snap[idx - 1], nxt = snap[idx];
            if (cur == 0 && nxt == 1) {
                if (offset) {
                    if (idx - offset_point < min_vally) fill(offset_point, idx, 1);
                    offset = false;
                }
            } else if (cur == 1 && nxt == 0) {
                offset = true;
                offset_point = idx;
            }
        }
    }
    if (min_hill > 0) {  // flatten hills shorter than min_hill (:33-48)
        snap.assign(out, out + n);
        bool onset = false;
        long onset_point = 0;
        for (int idx = 1; idx < n; ++idx) {
            const uint8_t cur = snap[idx - 1], nxt = snap[idx];
            if (cur == 0 && nxt == 1) {
                onset = true;
                onset_point = idx;
            } else if (cur == 1 && nxt == 0) {
                if (onset) {
                    if (idx - onset_point < min_hill) fill(onset_point, idx, 0);
                    onset = false;
                }
            }
        }
    }
    // extend both ends (:51-64).  The reference tests hang_before twice ("hang_before > 0 or hang_before > 0"):
    // hang_over alone never triggers this pass.  Kept.
    if (hang_before > 0 || hang_before > 0) {
        snap.assign(out, out + n);
        for (int idx = 1; idx < n; ++idx) {
            const uint8_t cur = snap[idx - 1], nxt = snap[idx];
            if (cur == 0 && nxt == 1) {
                if (idx < hang_before)
                    fill(0, idx, 1);
                else
                    fill((long)idx - hang_before, idx, 1);
            } else if (cur == 1 && nxt == 0) {
                if ((long)n - hang_over < idx)
                    fill(idx, n, 1);
                else
                    fill(idx, (long)idx + hang_over, 1);
            }
        }
    }
}

// vad/postprocessing/convert.py:6-24.  Returns the number of samples int((n-1)*hop + window); out may be null.
inline long frames_to_samples(const double* frames, int n, int sample_rate, double hop_ms, double window_ms, double* out) {
    const double hop = sample_rate * hop_ms / 1000, win = sample_rate * window_ms / 1000;
    const long num = (long)((n - 1) * hop + win);
    if (!out || num <= 0) return num > 0 ? num : 0;
    std::vector<double> counts((size_t)num, 0.0);
    std::fill(out, out + num, 0.0);
    double start = 0.0;  // float accumulation of the hop, as in the reference (:15-20)
    for (int f = 0; f < n; ++f) {
        long lo = (long)start, hi = (long)(start + win);
        lo = std::min(lo, num);
        hi = std::min(hi, num);
        for (long i = lo; i < hi; ++i) {
            out[i] += frames[f];
            counts[i] += 1.0;
        }
        start += hop;
    }
    for (long i = 0; i < num; ++i) out[i] /= (counts[i] == 0.0 ? 1.0 : counts[i]);
    return num;
}

// vad/postprocessing/convert.py:27-61.  Segment boundaries as SAMPLE INDICES (start index, end index) where the
// reference builds timedelta(seconds=index / sample_rate); end = sample_index - 1 at a 1 -> 0 switch, the last
// sample index if the signal ends in voice.  Only exact 1.0 / 0.0 values drive the state machine.
inline int samples_to_segments(const double* s, long n, long* starts, long* ends, int cap) {
    int cnt = 0;
    bool is_voice = false;
    long start = -1;
    for (long i = 0; i < n; ++i) {
        if (s[i] == 1.0 && !is_voice) {
            is_voice = true;
            start = i;
        }
        if (s[i] == 0.0 && is_voice) {
            is_voice = false;
            if (cnt < cap) {
                starts[cnt] = start;
                ends[cnt] = i - 1;
            }
            ++cnt;
            start = -1;
        }
    }
    if (n > 0 && is_voice) {
        if (cnt < cap) {
            starts[cnt] = start;
            ends[cnt] = n - 1;
        }
        ++cnt;
    }
    return cnt;
}

// vad/postprocessing/split.py:81-109 (recursive): break points inside probs[0:len)
inline void split_long_block(const double* p, long len, long max_samples, long base, std::vector<long>& breaks) {
    const long half = max_samples / 2;
    // trimmed = p[half : len-half]; np.argmin -> first minimum.  (-0 slice end cannot occur: half >= 1)
    long best = half;
    for (long i = half; i < len - half; ++i)
        if (p[i] < p[best]) best = i;
    const long bp = best;  // half + argmin(trimmed)
    if (bp > max_samples) split_long_block(p, bp, max_samples, base, breaks);
    breaks.push_back(base + bp);
    const long rlen = len - bp - 1;
    if (rlen > max_samples) split_long_block(p + bp + 1, rlen, max_samples, base + bp + 1, breaks);
}

// vad/postprocessing/split.py:26-78
inline void optimal_split(const double* pred, const double* probs, long n, long max_samples, double* out) {
    std::copy(pred, pred + n, out);
    bool is_voice = false;
    long start = -1;
    auto handle = [&](long s, long e) {
        if (e - s > max_samples) {
            std::vector<long> breaks;
            split_long_block(probs + s, e - s, max_samples, 0, breaks);
            for (long b : breaks) out[s + b] = 0.0;
        }
    };
    for (long i = 0; i < n; ++i) {
        if (pred[i] == 1.0 && !is_voice) {
            is_voice = true;
            start = i;
        }
        if (pred[i] == 0.0 && is_voice) {
            is_voice = false;
            if (start >= 0) handle(start, i);
            start = -1;
        }
    }
    if (n > 0 && start >= 0 && is_voice) handle(start, n);
}

}  // namespace post
}  // namespace savad
