// CIDEr-D reward on the device (float64), replacing the per-batch pure-Python scorer of the reference:
//   Utils.py:319-367 get_self_critical_reward -> ciderD.py:30-55 -> ciderD_scorer.py:17-32 (precook),
//   :127-206 (compute_cider).
// One workgroup of CD_NW waves per hypothesis (2B of them: B sampled, B greedy).  Hypotheses are at most T <= 60 tokens,
// so a hypothesis has at most 4T n-gram positions; threads work on positions in parallel (dedup, df lookup) and each wave
// matches the hypothesis against one of the image's cooked references at a time.  Every floating-point accumulation is
// performed by one lane in the reference's dict-insertion order (per reference by the wave's lane 0, across references
// by thread 0 in reference order), which makes the scores bit-identical to the reference's float64 results.
#include <math.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "icz_common.h"

namespace icz {

struct CiderD {
    int32_t* keys = nullptr;     // [cap,4]
    double* idf = nullptr;       // [cap]
    double* penalty = nullptr;   // [64]
    int64_t cap = 0;
    double default_idf = 0.0;
};

constexpr int CD_MAXT = 60;                 // max tokens per hypothesis
constexpr int CD_MAXP = 4 * CD_MAXT;        // max n-gram positions
constexpr int CD_NW = 8;                    // waves per hypothesis = references matched at a time

__host__ __device__ inline uint32_t ngram_hash(int a, int b, int c, int d) {
    uint32_t h = 2166136261u;
    h = (h ^ (uint32_t)a) * 16777619u;
    h = (h ^ (uint32_t)b) * 16777619u;
    h = (h ^ (uint32_t)c) * 16777619u;
    h = (h ^ (uint32_t)d) * 16777619u;
    h ^= h >> 15;
    return h;
}

struct CiderArgs {
    const int32_t* keys; const double* idf; const double* penalty; int64_t cap; double default_idf;
    const int64_t* gen; const int64_t* greedy; int B, T;
    const int32_t* img_slot;     // [B] slot of image b in the reference store, or null = b (the arrays below are the batch's own)
    const int32_t* img_ref_ptr; const int32_t* ref_ent_ptr; const int32_t* ent_key; const int32_t* ent_order;
    const double* ent_w; const double* ref_norm; const int32_t* ref_len;
    double* scores;      // [2B]
};

__global__ __launch_bounds__(64 * CD_NW) void ciderd_kernel(CiderArgs a) {
    __shared__ int tok[CD_MAXT];
    __shared__ int pkey[CD_MAXP][4];
    __shared__ int porder[CD_MAXP];       // 1..4
    __shared__ int pcount[CD_MAXP];       // occurrences if this position is the first occurrence, else 0
    __shared__ double pw[CD_MAXP];        // tf-idf weight of the n-gram first seen at this position
    __shared__ double pmatch[CD_NW][CD_MAXP];    // per wave: weight of the same n-gram in the wave's current reference (0 if absent)
    __shared__ double rval[CD_NW][4];     // per wave: the reference's contribution to the four n-gram orders
    constexpr int NT = 64 * CD_NW;
    const int hyp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = hyp % a.B;
    const bool is_greedy = hyp >= a.B;
    const int64_t* ids = (is_greedy ? a.greedy : a.gen) + (size_t)b * a.T;
    // ---- sentence length (Utils.py:337-356)
    int len;
    if (is_greedy) {
        len = a.T;
        for (int i = 0; i < a.T; ++i)
            if (ids[i] == 2) { len = i; break; }
    } else {
        int end = 0;
        for (int e = a.T - 1; e >= 0; --e) {
            end = e;
            if (ids[e] != 0) break;
        }
        len = end + 1;
    }
    for (int i = tid; i < len; i += NT) tok[i] = (int)ids[i];
    __syncthreads();
    // ---- n-gram positions in precook order: k = 1..4, i = 0..len-k
    int npos = 0, start[5];
    for (int k = 1; k <= 4; ++k) { start[k] = npos; npos += (len - k + 1 > 0) ? (len - k + 1) : 0; }
    for (int p = tid; p < npos; p += NT) {
        int k = 4;
        while (k > 1 && p < start[k]) --k;
        const int i = p - start[k];
        porder[p] = k;
        for (int j = 0; j < 4; ++j) pkey[p][j] = (j < k) ? tok[i + j] : -1;
    }
    __syncthreads();
    // ---- dedup: a position is "first" if no earlier position holds the same n-gram; count = #occurrences
    for (int p = tid; p < npos; p += NT) {
        const int k = porder[p];
        bool first = true;
        int cnt = 0;
        const int s0 = start[k], s1 = s0 + (len - k + 1);
        for (int q = s0; q < s1; ++q) {
            const bool same = pkey[q][0] == pkey[p][0] && pkey[q][1] == pkey[p][1] && pkey[q][2] == pkey[p][2] && pkey[q][3] == pkey[p][3];
            if (same) { ++cnt; if (q < p) first = false; }
        }
        pcount[p] = first ? cnt : 0;
        double w = 0.0;
        if (first) {
            // document-frequency lookup (open addressing, linear probing)
            double idfv = a.default_idf;
            uint32_t h = ngram_hash(pkey[p][0], pkey[p][1], pkey[p][2], pkey[p][3]);
            for (int64_t probe = 0; probe < a.cap; ++probe) {
                const int64_t s = (int64_t)((h + (uint32_t)probe) & (uint32_t)(a.cap - 1));
                const int32_t* kk = a.keys + s * 4;
                if (kk[0] == -1) break;   // empty slot (token ids are >= 0, so key[0] == -1 marks empty)
                if (kk[0] == pkey[p][0] && kk[1] == pkey[p][1] && kk[2] == pkey[p][2] && kk[3] == pkey[p][3]) { idfv = a.idf[s]; break; }
            }
            w = (double)cnt * idfv;      // float(term_freq) * (ref_len - df)   (ciderD_scorer.py:145)
        }
        pw[p] = w;
    }
    __syncthreads();
    // ---- hypothesis norms and length (thread 0, insertion order)   (:146-152)
    __shared__ double hnorm[4];
    __shared__ int hlen;
    if (tid == 0) {
        double nn[4] = {0.0, 0.0, 0.0, 0.0};
        int l2 = 0;
        for (int p = 0; p < npos; ++p)
            if (pcount[p]) {
                nn[porder[p] - 1] += pw[p] * pw[p];
                if (porder[p] == 2) l2 += pcount[p];
            }
        for (int n = 0; n < 4; ++n) hnorm[n] = sqrt(nn[n]);
        hlen = l2;
    }
    __syncthreads();
    // ---- references of this image, CD_NW at a time: wave w takes reference rc + w
    const int slot = a.img_slot ? a.img_slot[b] : b;
    const int r0 = a.img_ref_ptr[slot], r1 = a.img_ref_ptr[slot + 1];
    double score[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rc = r0; rc < r1; rc += CD_NW) {
        const int r = rc + wave;
        if (r < r1) {
            const int e0 = a.ref_ent_ptr[r], e1 = a.ref_ent_ptr[r + 1];
            for (int p = lane; p < npos; p += 64) {
                double m = 0.0;
                if (pcount[p]) {
                    for (int e = e0; e < e1; ++e) {
                        const int32_t* kk = a.ent_key + (size_t)e * 4;
                        if (a.ent_order[e] == porder[p] && kk[0] == pkey[p][0] && kk[1] == pkey[p][1] && kk[2] == pkey[p][2] && kk[3] == pkey[p][3]) {
                            m = a.ent_w[e];
                            break;
                        }
                    }
                }
                pmatch[wave][p] = m;
            }
        }
        __syncthreads();
        if (r < r1 && lane == 0) {
            double val[4] = {0.0, 0.0, 0.0, 0.0};
            for (int p = 0; p < npos; ++p)
                if (pcount[p]) {
                    const double h = pw[p], rr = pmatch[wave][p];
                    val[porder[p] - 1] += (h < rr ? h : rr) * rr;       // min(vec_hyp, vec_ref) * vec_ref  (:172-175)
                }
            int d = hlen - a.ref_len[r];
            if (d < 0) d = -d;
            const double pen = a.penalty[d > 63 ? 63 : d];
            for (int n = 0; n < 4; ++n) {
                const double nr = a.ref_norm[(size_t)r * 4 + n];
                if (hnorm[n] != 0.0 && nr != 0.0) val[n] /= (hnorm[n] * nr);
                val[n] *= pen;
                rval[wave][n] = val[n];
            }
        }
        __syncthreads();
        if (tid == 0) {
            const int nr_ = r1 - rc < CD_NW ? r1 - rc : CD_NW;
            for (int w = 0; w < nr_; ++w)
                for (int n = 0; n < 4; ++n) score[n] += rval[w][n];       // reference order, as the scorer's loop (:186-196)
        }
        __syncthreads();
    }
    if (tid == 0) {
        double s = score[0];
        s += score[1]; s += score[2]; s += score[3];
        s = s / 4.0;                      // np.mean over n
        s /= (double)(r1 - r0);           // / len(refs)
        s *= 10.0;
        a.scores[hyp] = s;
    }
}

__global__ void ciderd_reward_kernel(const double* __restrict__ scores, int B, int T, float* __restrict__ reward) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    int b = i / T;
    reward[i] = (float)(scores[b] - scores[B + b]);
}

}  // namespace icz

using namespace icz;
extern "C" {

int icz_ciderd_create(const int32_t* df_keys, const double* df_idf, int64_t cap, double default_idf,
                      const double* penalty, icz_ciderd_t** out) {
    ICZ_REQUIRE(df_keys && df_idf && penalty && out, "icz_ciderd_create: null argument");
    ICZ_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0, "icz_ciderd_create: cap must be a power of two");
    CiderD* c = new CiderD();
    c->cap = cap;
    c->default_idf = default_idf;
    c->keys = const_cast<int32_t*>(df_keys);
    c->idf = const_cast<double*>(df_idf);
    c->penalty = const_cast<double*>(penalty);
    *out = reinterpret_cast<icz_ciderd_t*>(c);
    return ICZ_OK;
}

int icz_ciderd_destroy(icz_ciderd_t* h) {
    delete reinterpret_cast<CiderD*>(h);
    return ICZ_OK;
}

static int ciderd_reward_impl(icz_ciderd_t* h, const int64_t* gen, const int64_t* greedy, int32_t B, int32_t T, const int32_t* img_slot,
                              const int32_t* img_ref_ptr, const int32_t* ref_ent_ptr, const int32_t* ent_key,
                              const int32_t* ent_order, const double* ent_w, const double* ref_norm, const int32_t* ref_len,
                              float* reward_out, double* scores_out, void* stream) {
    ICZ_REQUIRE(h && gen && greedy && img_ref_ptr && ref_ent_ptr && ent_key && ent_order && ent_w && ref_norm && ref_len,
                "icz_ciderd_reward: null argument");
    ICZ_REQUIRE(scores_out, "icz_ciderd_reward: scores_out (2B float64 scratch) is required");
    ICZ_REQUIRE(B > 0 && T > 0 && T <= CD_MAXT, "icz_ciderd_reward: T=%d out of range 1..%d", T, CD_MAXT);
    CiderD* c = reinterpret_cast<CiderD*>(h);
    CiderArgs a = {c->keys, c->idf, c->penalty, c->cap, c->default_idf, gen, greedy, B, T, img_slot,
                   img_ref_ptr, ref_ent_ptr, ent_key, ent_order, ent_w, ref_norm, ref_len, scores_out};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ciderd_kernel, dim3(2 * B), dim3(64 * CD_NW), 0, st, a);
    if (reward_out) hipLaunchKernelGGL(ciderd_reward_kernel, dim3(cdiv(B * T, 256)), dim3(256), 0, st, scores_out, B, T, reward_out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// Host side of the reference store: "cooking" references (ciderD_scorer.py:17-32 precook + :128-153 counts2vec) from token
// ids -- the reference does this in Python dict loops for every batch; here once per image, ~100x faster than the Python
// restatement in ciderd.py (which stays as the checker of this function, tests/test_cpu_abi_and_host.py).  Pure host code:
// no device memory is touched.  Entries of a reference come out in the scorer's dict-insertion order (order k ascending,
// first occurrence ascending), weights / norms in the scorer's float64 arithmetic (pow(w, 2) summed in that order, sqrt).
int icz_ciderd_cook_host(const int32_t* df_keys_host, const double* df_idf_host, int64_t cap, double default_idf,
                         const int32_t* tokens, const int32_t* ref_tok_ptr, int32_t n_refs, int64_t max_ent,
                         int32_t* ent_key_out, int32_t* ent_order_out, double* ent_w_out, int32_t* ref_ent_ptr_out,
                         double* ref_norm_out, int32_t* ref_len_out, int64_t* n_ent_out) {
    ICZ_REQUIRE(df_keys_host && df_idf_host && tokens && ref_tok_ptr && ent_key_out && ent_order_out && ent_w_out && ref_ent_ptr_out &&
                ref_norm_out && ref_len_out && n_ent_out, "icz_ciderd_cook_host: null argument");
    ICZ_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0 && n_refs >= 0, "icz_ciderd_cook_host: bad table size / reference count");
    int64_t ne = 0;
    ref_ent_ptr_out[0] = 0;
    std::vector<int32_t> cnt;
    for (int32_t r = 0; r < n_refs; ++r) {
        const int32_t* t = tokens + ref_tok_ptr[r];
        const int L = ref_tok_ptr[r + 1] - ref_tok_ptr[r];
        const int64_t e0 = ne;
        cnt.clear();
        for (int k = 1; k <= 4; ++k) {
            const int64_t k0 = ne;                       // entries of order k start here
            for (int i = 0; i + k <= L; ++i) {
                int32_t key[4] = {-1, -1, -1, -1};
                for (int j = 0; j < k; ++j) key[j] = t[i + j];
                int64_t hit = -1;
                for (int64_t e = k0; e < ne; ++e) {
                    const int32_t* kk = ent_key_out + e * 4;
                    if (kk[0] == key[0] && kk[1] == key[1] && kk[2] == key[2] && kk[3] == key[3]) { hit = e; break; }
                }
                if (hit >= 0) { ++cnt[(size_t)(hit - e0)]; continue; }
                ICZ_REQUIRE(ne < max_ent, "icz_ciderd_cook_host: more than %lld n-gram entries", (long long)max_ent);
                int32_t* o = ent_key_out + ne * 4;
                o[0] = key[0]; o[1] = key[1]; o[2] = key[2]; o[3] = key[3];
                ent_order_out[ne] = k;
                cnt.push_back(1);
                ++ne;
            }
        }
        double norm[4] = {0.0, 0.0, 0.0, 0.0};
        int32_t len2 = 0;
        for (int64_t e = e0; e < ne; ++e) {
            const int32_t* kk = ent_key_out + e * 4;
            double idfv = default_idf;
            const uint32_t hsh = ngram_hash(kk[0], kk[1], kk[2], kk[3]);
            for (int64_t probe = 0; probe < cap; ++probe) {
                const int64_t s = (int64_t)((hsh + (uint32_t)probe) & (uint32_t)(cap - 1));
                const int32_t* tk = df_keys_host + s * 4;
                if (tk[0] == -1) break;
                if (tk[0] == kk[0] && tk[1] == kk[1] && tk[2] == kk[2] && tk[3] == kk[3]) { idfv = df_idf_host[s]; break; }
            }
            const double w = (double)cnt[(size_t)(e - e0)] * idfv;     // float(term_freq) * (ref_len - df)   (ciderD_scorer.py:145)
            ent_w_out[e] = w;
            norm[ent_order_out[e] - 1] += pow(w, 2.0);                   // norm[n] += pow(vec[n][ngram], 2)   (:147)
            if (ent_order_out[e] == 2) len2 += cnt[(size_t)(e - e0)];
        }
        for (int n = 0; n < 4; ++n) ref_norm_out[(size_t)r * 4 + n] = sqrt(norm[n]);
        ref_len_out[r] = len2;
        ref_ent_ptr_out[r + 1] = (int32_t)ne;
    }
    *n_ent_out = ne;
    return ICZ_OK;
}

// Word -> id map of the scorer on the host side of the library: the caption vocabulary plus private ids (>= V, in order of first
// appearance) for words outside it, which the references and the document-frequency table may contain (they count in the
// reference norms and never match a hypothesis).  One owner for those private ids: the Python cooker asks here too
// (icz_ciderd_vocab_oov_id), so that references tokenised in C++ (icz_ciderd_cook_text) and n-grams keyed in Python agree.
struct CiderVocab {
    std::unordered_map<std::string, int32_t> base, ext;
    int32_t V = 0;
    std::mutex mu;
    int32_t id_of(const std::string& w) {
        auto it = base.find(w);
        if (it != base.end()) return it->second;
        std::lock_guard<std::mutex> lk(mu);
        auto e = ext.find(w);
        if (e != ext.end()) return e->second;
        const int32_t id = V + (int32_t)ext.size();
        ext.emplace(w, id);
        return id;
    }
};

int icz_ciderd_vocab_create(const char* words, const int64_t* word_off, const int32_t* word_id, int32_t n_words, int32_t V,
                            icz_ciderd_vocab_t** out) {
    ICZ_REQUIRE(words && word_off && word_id && n_words >= 0 && V > 0 && out, "icz_ciderd_vocab_create: bad arguments");
    CiderVocab* v = new CiderVocab();
    v->V = V;
    v->base.reserve((size_t)n_words * 2);
    for (int32_t i = 0; i < n_words; ++i) v->base.emplace(std::string(words + word_off[i], (size_t)(word_off[i + 1] - word_off[i])), word_id[i]);
    *out = reinterpret_cast<icz_ciderd_vocab_t*>(v);
    return ICZ_OK;
}
int icz_ciderd_vocab_destroy(icz_ciderd_vocab_t* v) {
    delete reinterpret_cast<CiderVocab*>(v);
    return ICZ_OK;
}
int icz_ciderd_vocab_oov_id(icz_ciderd_vocab_t* v, const char* word, int32_t len, int32_t* id_out) {
    ICZ_REQUIRE(v && word && len >= 0 && id_out, "icz_ciderd_vocab_oov_id: bad arguments");
    *id_out = reinterpret_cast<CiderVocab*>(v)->id_of(std::string(word, (size_t)len));
    return ICZ_OK;
}

// icz_ciderd_cook_host with the tokenisation in front of it: `text` holds n_refs references separated by '\n', words separated
// by runs of ASCII whitespace (str.split() of an ASCII string: the caller checks that the text is ASCII and has exactly
// n_refs - 1 newlines).  Everything between the caller's join() and the flat arrays happens here, outside the interpreter lock:
// the loader thread that cooks the next batch's references no longer competes with the training thread's kernel launches.
int icz_ciderd_cook_text(icz_ciderd_vocab_t* vocab, const int32_t* df_keys_host, const double* df_idf_host, int64_t cap, double default_idf,
                         const char* text, int64_t text_len, int32_t n_refs, int64_t max_ent,
                         int32_t* ent_key_out, int32_t* ent_order_out, double* ent_w_out, int32_t* ref_ent_ptr_out,
                         double* ref_norm_out, int32_t* ref_len_out, int64_t* n_ent_out) {
    ICZ_REQUIRE(vocab && text && text_len >= 0 && n_refs >= 0, "icz_ciderd_cook_text: bad arguments");
    CiderVocab* v = reinterpret_cast<CiderVocab*>(vocab);
    std::vector<int32_t> tok, ptr;
    tok.reserve((size_t)text_len / 2 + 8);
    ptr.reserve((size_t)n_refs + 1);
    ptr.push_back(0);
    std::string w;
    int64_t i = 0;
    int32_t refs = 0;
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || (c >= 0x1c && c <= 0x1f); };    // str.split()'s ASCII whitespace minus '\n'
    while (refs < n_refs) {
        while (i < text_len && text[i] != '\n') {
            while (i < text_len && is_space(text[i])) ++i;
            const int64_t b = i;
            while (i < text_len && text[i] != '\n' && !is_space(text[i])) ++i;
            if (i > b) { w.assign(text + b, (size_t)(i - b)); tok.push_back(v->id_of(w)); }
        }
        ++i;                                     // the newline (or one past the end after the last reference)
        ptr.push_back((int32_t)tok.size());
        ++refs;
    }
    ICZ_REQUIRE(i >= text_len, "icz_ciderd_cook_text: text holds more than %d references", n_refs);
    if (tok.empty()) tok.push_back(0);           // keep the pointer valid
    return icz_ciderd_cook_host(df_keys_host, df_idf_host, cap, default_idf, tok.data(), ptr.data(), n_refs, max_ent, ent_key_out, ent_order_out,
                                ent_w_out, ref_ent_ptr_out, ref_norm_out, ref_len_out, n_ent_out);
}

int icz_ciderd_reward(icz_ciderd_t* h, const int64_t* gen, const int64_t* greedy, int32_t B, int32_t T,
                      const int32_t* img_ref_ptr, const int32_t* ref_ent_ptr, const int32_t* ent_key,
                      const int32_t* ent_order, const double* ent_w, const double* ref_norm, const int32_t* ref_len,
                      float* reward_out, double* scores_out, void* stream) {
    return ciderd_reward_impl(h, gen, greedy, B, T, nullptr, img_ref_ptr, ref_ent_ptr, ent_key, ent_order, ent_w, ref_norm, ref_len,
                              reward_out, scores_out, stream);
}

int icz_ciderd_reward_indexed(icz_ciderd_t* h, const int64_t* gen, const int64_t* greedy, int32_t B, int32_t T,
                              const int32_t* img_slot, const int32_t* img_ref_ptr, const int32_t* ref_ent_ptr,
                              const int32_t* ent_key, const int32_t* ent_order, const double* ent_w, const double* ref_norm,
                              const int32_t* ref_len, float* reward_out, double* scores_out, void* stream) {
    ICZ_REQUIRE(img_slot, "icz_ciderd_reward_indexed: null img_slot");
    return ciderd_reward_impl(h, gen, greedy, B, T, img_slot, img_ref_ptr, ref_ent_ptr, ent_key, ent_order, ent_w, ref_norm, ref_len,
                              reward_out, scores_out, stream);
}

}  // extern "C"
