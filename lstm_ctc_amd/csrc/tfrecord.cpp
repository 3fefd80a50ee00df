// tfrecord.cpp — HOST side of the input path: TFRecord framing (with CRC-32C verification) and tf.train.SequenceExample
// decoding, splice and subsample fused into the copy.  No GPU code; part of liblstm_ctc_hip.so so that the loader's
// worker threads run native code with the GIL released.
//
// What it replaces in the reference: tf.data.TFRecordDataset + tf.parse_single_sequence_example with
// FixedLenSequenceFeature([input_dim], float32) / FixedLenSequenceFeature([], int64) and the _splice / _subsample graph ops
// (nnet/tfrecord.py:28-51, 94-125).  Formats: SURVEY.md Appendix C.
//
//   record    = uint64 length | uint32 masked_crc32c(length bytes) | payload | uint32 masked_crc32c(payload)
//   payload   = SequenceExample { 1: Features context (skipped), 2: FeatureLists { 1: map<string, FeatureList> } }
//   FeatureList { 1: repeated Feature }, Feature { oneof 1: BytesList, 2: FloatList, 3: Int64List }, *List { 1: repeated }
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <stdio.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lstm_ctc_hip.h"

void lc_set_error(const char *fmt, ...);

// ---------------------------------------------------------------------------------------------------- CRC-32C
namespace {
uint32_t g_tab[8][256];
bool g_tab_ready = false;
void crc_tables()
{
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        g_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_tab[t][i] = (g_tab[t - 1][i] >> 8) ^ g_tab[0][g_tab[t - 1][i] & 0xff];
    g_tab_ready = true;
}
struct CrcInit { CrcInit() { crc_tables(); } } g_crc_init;

uint32_t crc_sw(uint32_t c, const uint8_t *p, size_t n)
{
    if (!g_tab_ready) crc_tables();
    while (n && ((uintptr_t)p & 7)) { c = g_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = g_tab[7][w & 0xff] ^ g_tab[6][(w >> 8) & 0xff] ^ g_tab[5][(w >> 16) & 0xff] ^ g_tab[4][(w >> 24) & 0xff] ^
            g_tab[3][(w >> 32) & 0xff] ^ g_tab[2][(w >> 40) & 0xff] ^ g_tab[1][(w >> 48) & 0xff] ^ g_tab[0][w >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = g_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c;
}
#if defined(__x86_64__)
// three independent crc32 chains per iteration would need a carry-less-multiply merge; one chain at 8 bytes per 3 cycles
// (~5 GB/s) is already 10x what a GPU step consumes
__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t c, const uint8_t *p, size_t n)
{
    uint64_t c64 = c;
    while (n && ((uintptr_t)p & 7)) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        c64 = __builtin_ia32_crc32di(c64, w);
        p += 8; n -= 8;
    }
    while (n--) c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
    return (uint32_t)c64;
}
#endif
uint32_t crc32c(const uint8_t *p, size_t n)
{
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc_hw(0xffffffffu, p, n) ^ 0xffffffffu;
#endif
    return crc_sw(0xffffffffu, p, n) ^ 0xffffffffu;
}
inline uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---------------------------------------------------------------------------------------------------- wire format
struct Span {
    const uint8_t *p, *end;
    bool empty() const { return p >= end; }
};
bool varint(Span &s, uint64_t &v)
{
    v = 0;
    for (int shift = 0; shift < 64 && s.p < s.end; shift += 7) {
        const uint8_t b = *s.p++;
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}
// next field of a message: number, wire type, and for length-delimited fields the sub-span; scalar fields are consumed
struct Field {
    uint32_t num;
    int wt;
    uint64_t val;      // wt 0
    Span sub;          // wt 2 (and the 4 / 8 raw bytes of wt 5 / 1)
};
bool next_field(Span &s, Field &f)
{
    uint64_t key;
    if (!varint(s, key)) return false;
    f.num = (uint32_t)(key >> 3);
    f.wt = (int)(key & 7);
    switch (f.wt) {
    case 0: return varint(s, f.val);
    case 1: if (s.end - s.p < 8) return false; f.sub = {s.p, s.p + 8}; s.p += 8; return true;
    case 5: if (s.end - s.p < 4) return false; f.sub = {s.p, s.p + 4}; s.p += 4; return true;
    case 2: {
        uint64_t n;
        if (!varint(s, n) || n > (uint64_t)(s.end - s.p)) return false;
        f.sub = {s.p, s.p + n};
        s.p += n;
        return true;
    }
    default: return false;      // groups (3 / 4) do not occur in tf.train.*
    }
}

struct Lists {
    Span input{nullptr, nullptr}, target{nullptr, nullptr};      // the FeatureList messages
    bool has_input = false, has_target = false;
};
// SequenceExample -> the two FeatureList spans (map entries in any order; a repeated key: the last one wins)
bool find_lists(Span ex, Lists &out)
{
    Field f;
    while (!ex.empty()) {
        if (!next_field(ex, f)) return false;
        if (f.num != 2 || f.wt != 2) continue;             // 1 = context
        Span fls = f.sub;
        Field e;
        while (!fls.empty()) {
            if (!next_field(fls, e)) return false;
            if (e.num != 1 || e.wt != 2) continue;
            Span entry = e.sub, key{nullptr, nullptr}, val{nullptr, nullptr};
            Field kv;
            while (!entry.empty()) {
                if (!next_field(entry, kv)) return false;
                if (kv.wt != 2) continue;
                if (kv.num == 1) key = kv.sub;
                else if (kv.num == 2) val = kv.sub;
            }
            const size_t kl = (size_t)(key.end - key.p);
            if (kl == 10 && memcmp(key.p, "nnet_input", 10) == 0) { out.input = val; out.has_input = true; }
            else if (kl == 11 && memcmp(key.p, "nnet_target", 11) == 0) { out.target = val; out.has_target = true; }
        }
    }
    return true;
}
// One Feature of nnet_input: where its floats are.  packed -> `data` points at dim*4 contiguous bytes inside the record.
struct Row {
    const uint8_t *data;      // packed payload, or nullptr when the values are stored one by one
    Span list;                // the FloatList message (for the unpacked walk)
    int64_t n;                // number of floats
};
bool float_row(Span feat, Row &r)
{
    r.data = nullptr; r.n = 0; r.list = {nullptr, nullptr};
    Field f;
    while (!feat.empty()) {
        if (!next_field(feat, f)) return false;
        if (f.num != 2 || f.wt != 2) continue;             // float_list
        r.list = f.sub;
        Span fl = f.sub;
        Field v;
        int pieces = 0;
        int64_t n = 0;
        const uint8_t *first = nullptr;
        while (!fl.empty()) {
            if (!next_field(fl, v)) return false;
            if (v.num != 1) continue;
            if (v.wt == 2) {
                const size_t nb = (size_t)(v.sub.end - v.sub.p);
                if (nb % 4) return false;
                if (!pieces) first = v.sub.p;
                n += (int64_t)(nb / 4);
                ++pieces;
            } else if (v.wt == 5) { n += 1; pieces += 2; }   // unpacked: never the single-piece fast path
            else return false;
        }
        r.n = n;
        r.data = pieces == 1 ? first : nullptr;
    }
    return true;
}
void copy_row(const Row &r, float *dst)
{
    if (r.data) { memcpy(dst, r.data, (size_t)r.n * 4); return; }
    Span fl = r.list;
    Field v{};
    while (!fl.empty() && next_field(fl, v)) {
        if (v.num != 1) continue;
        const size_t nb = (size_t)(v.sub.end - v.sub.p);
        memcpy(dst, v.sub.p, nb);
        dst += nb / 4;
    }
}
// One Feature of nnet_target: exactly one int64 (FixedLenSequenceFeature(shape=[]))
bool int64_scalar(Span feat, int64_t &out)
{
    Field f;
    int64_t n = 0;
    while (!feat.empty()) {
        if (!next_field(feat, f)) return false;
        if (f.num != 3 || f.wt != 2) continue;             // int64_list
        Span il = f.sub;
        Field v;
        while (!il.empty()) {
            if (!next_field(il, v)) return false;
            if (v.num != 1) continue;
            if (v.wt == 0) { out = (int64_t)v.val; ++n; }
            else if (v.wt == 2) {
                Span pk = v.sub;
                uint64_t x;
                while (!pk.empty()) {
                    if (!varint(pk, x)) return false;
                    out = (int64_t)x;
                    ++n;
                }
            } else return false;
        }
    }
    return n == 1;
}
// first record of the file image -> payload span (CRCs checked on request)
int first_record(const uint8_t *file, size_t nbytes, int verify_crc, Span &payload)
{
    if (nbytes < 16) { lc_set_error("tfrecord: file of %zu bytes holds no record", nbytes); return LC_EINVAL; }
    uint64_t len;
    uint32_t crc;
    memcpy(&len, file, 8);
    memcpy(&crc, file + 8, 4);
    if (verify_crc && masked(crc32c(file, 8)) != crc) {
        lc_set_error("tfrecord: corrupted record header (length CRC mismatch)");
        return LC_EINVAL;
    }
    if (len > nbytes - 16) { lc_set_error("tfrecord: truncated record (%llu payload bytes, %zu in file)",
                                          (unsigned long long)len, nbytes - 16); return LC_EINVAL; }
    payload = {file + 12, file + 12 + len};
    if (verify_crc) {
        memcpy(&crc, file + 12 + len, 4);
        if (masked(crc32c(payload.p, (size_t)len)) != crc) {
            lc_set_error("tfrecord: corrupted record (payload CRC mismatch)");
            return LC_EINVAL;
        }
    }
    // tf.data.TFRecordDataset (nnet/tfrecord.py:122) yields EVERY record of a file; this loader maps one file to one
    // utterance (what bin/convert-to-tfrecords.py writes).  Bytes behind the first record - further records, garbage - would
    // be utterances silently dropped: refuse them.
    if ((size_t)len + 16 != nbytes) {
        lc_set_error("tfrecord: %zu bytes behind the first record (further records or trailing bytes): one SequenceExample "
                     "per file is supported", nbytes - 16 - (size_t)len);
        return LC_EINVAL;
    }
    return LC_OK;
}
} // namespace

extern "C" uint32_t lc_crc32c(const void *data, size_t nbytes) { return crc32c((const uint8_t *)data, nbytes); }

extern "C" int lc_tfrecord_inspect(const void *file, size_t nbytes, int verify_crc, lc_seqex_info_t *info)
{
    if (!file || !info) { lc_set_error("lc_tfrecord_inspect: bad argument"); return LC_EINVAL; }
    Span payload;
    const int rc = first_record((const uint8_t *)file, nbytes, verify_crc, payload);
    if (rc != LC_OK) return rc;
    Lists ls;
    if (!find_lists(payload, ls)) { lc_set_error("tfrecord: malformed SequenceExample"); return LC_EINVAL; }
    info->num_frames = 0; info->dim = 0; info->num_labels = 0;
    info->has_input = ls.has_input; info->has_target = ls.has_target;
    Field f;
    Span in = ls.input;
    while (ls.has_input && !in.empty()) {
        if (!next_field(in, f)) { lc_set_error("tfrecord: malformed nnet_input list"); return LC_EINVAL; }
        if (f.num != 1 || f.wt != 2) continue;
        Row r;
        if (!float_row(f.sub, r)) { lc_set_error("tfrecord: malformed nnet_input feature"); return LC_EINVAL; }
        if (info->num_frames == 0) info->dim = (int32_t)r.n;
        else if (r.n != info->dim) {
            lc_set_error("tfrecord: nnet_input frame %lld has %lld values, frame 0 has %d",
                         (long long)info->num_frames, (long long)r.n, info->dim);
            return LC_EINVAL;
        }
        ++info->num_frames;
    }
    Span tg = ls.target;
    while (ls.has_target && !tg.empty()) {
        if (!next_field(tg, f)) { lc_set_error("tfrecord: malformed nnet_target list"); return LC_EINVAL; }
        if (f.num == 1 && f.wt == 2) ++info->num_labels;
    }
    return LC_OK;
}

extern "C" int lc_tfrecord_decode(const void *file, size_t nbytes, int dim, int left_context, int right_context,
                                  int subsample, float *x, size_t x_row_stride, int64_t max_rows, int64_t *labels,
                                  int64_t max_labels)
{
    if (!file || dim <= 0 || left_context < 0 || right_context < 0 || subsample < 0) {
        lc_set_error("lc_tfrecord_decode: bad argument");
        return LC_EINVAL;
    }
    Span payload;
    const int rc = first_record((const uint8_t *)file, nbytes, 0, payload);      // CRCs were checked by inspect
    if (rc != LC_OK) return rc;
    Lists ls;
    if (!find_lists(payload, ls)) { lc_set_error("tfrecord: malformed SequenceExample"); return LC_EINVAL; }
    Field f;
    if (x) {
        std::vector<Row> rows;
        Span in = ls.input;
        while (ls.has_input && !in.empty()) {
            if (!next_field(in, f)) { lc_set_error("tfrecord: malformed nnet_input list"); return LC_EINVAL; }
            if (f.num != 1 || f.wt != 2) continue;
            Row r;
            if (!float_row(f.sub, r) || r.n != dim) {
                lc_set_error("tfrecord: nnet_input frame %zu does not hold %d floats", rows.size(), dim);
                return LC_EINVAL;
            }
            rows.push_back(r);
        }
        const int64_t T = (int64_t)rows.size();
        const int64_t Tout = subsample > 0 ? T / subsample : T;        // tf.range(T / factor) * factor
        const int step = subsample > 0 ? subsample : 1;
        const int ctx = left_context + right_context + 1;
        if (Tout > max_rows || x_row_stride < (size_t)dim * ctx) {
            lc_set_error("lc_tfrecord_decode: output holds %lld rows of %zu floats, %lld x %d needed", (long long)max_rows,
                         x_row_stride, (long long)Tout, dim * ctx);
            return LC_EINVAL;
        }
        for (int64_t j = 0; j < Tout; ++j) {
            float *dst = x + (size_t)j * x_row_stride;
            const int64_t t = j * step;
            for (int c = 0; c < ctx; ++c) {                  // row t of the spliced matrix = [x[t-l] .. x[t] .. x[t+r]], edges
                int64_t s = t - left_context + c;            // replicated (tfrecord.py:28-40)
                s = s < 0 ? 0 : (s >= T ? T - 1 : s);
                copy_row(rows[(size_t)s], dst + (size_t)c * dim);
            }
        }
    }
    if (labels) {
        Span tg = ls.target;
        int64_t n = 0;
        while (ls.has_target && !tg.empty()) {
            if (!next_field(tg, f)) { lc_set_error("tfrecord: malformed nnet_target list"); return LC_EINVAL; }
            if (f.num != 1 || f.wt != 2) continue;
            int64_t v;
            if (!int64_scalar(f.sub, v)) { lc_set_error("tfrecord: nnet_target step %lld is not one int64", (long long)n); return LC_EINVAL; }
            if (n >= max_labels) { lc_set_error("lc_tfrecord_decode: more than %lld labels", (long long)max_labels); return LC_EINVAL; }
            labels[n++] = v;
        }
    }
    return LC_OK;
}

// ---------------------------------------------------------------------------------------------------- a whole batch
// The per-utterance calls above are fine from one thread, but a Python thread pool around 0.2 ms calls spends its time
// handing the GIL back and forth (measured: 8 threads SLOWER than one).  So a batch is two native calls, each fanning out
// over std::threads: open (read every file, check CRCs, count) and - once the caller knows the padded shape - decode
// (copy with splice / subsample into the utterance's rows, write the padding too).
struct lc_batch_reader {
    std::vector<std::string> paths;
    std::vector<std::vector<uint8_t>> raw;
    int dim = 0;
};
namespace {
template <class F>
int run_parallel(int n, int nthreads, F &&work, std::string &err)
{
    std::atomic<int> next{0}, rc{LC_OK};
    std::vector<std::string> errs((size_t)(nthreads > 0 ? nthreads : 1));
    auto body = [&](int tid) {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || rc.load() != LC_OK) return;
            const int r = work(i);
            if (r != LC_OK) {
                errs[(size_t)tid] = lc_last_error();       // thread-local message of THIS worker
                rc.store(r);
                return;
            }
        }
    };
    nthreads = nthreads < 1 ? 1 : (nthreads > n ? (n > 0 ? n : 1) : nthreads);
    errs.resize((size_t)nthreads);
    std::vector<std::thread> ts;
    for (int t = 1; t < nthreads; ++t) ts.emplace_back(body, t);
    body(0);
    for (auto &t : ts) t.join();
    for (auto &e : errs)
        if (!e.empty()) { err = e; break; }
    return rc.load();
}
bool read_file(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long n = ok ? ftell(f) : -1;
    ok = ok && n >= 0 && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        out.resize((size_t)n);
        ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    }
    fclose(f);
    return ok;
}
} // namespace

extern "C" int lc_batch_open(const char *const *paths, int n, int verify_crc, int expect_dim, int nthreads,
                             lc_batch_reader_t **reader, int64_t *num_frames, int64_t *num_labels)
{
    if (!paths || n < 0 || !reader || !num_frames || !num_labels) { lc_set_error("lc_batch_open: bad argument"); return LC_EINVAL; }
    lc_batch_reader *r = new lc_batch_reader;
    r->paths.assign(paths, paths + n);
    r->raw.resize((size_t)n);
    r->dim = expect_dim;
    std::string err;
    const int rc = run_parallel(n, nthreads, [&](int i) -> int {
        if (!read_file(r->paths[(size_t)i], r->raw[(size_t)i])) {
            lc_set_error("%s: cannot read", r->paths[(size_t)i].c_str());
            return LC_EINVAL;
        }
        lc_seqex_info_t info;
        const int rc1 = lc_tfrecord_inspect(r->raw[(size_t)i].data(), r->raw[(size_t)i].size(), verify_crc, &info);
        if (rc1 != LC_OK) {
            const std::string why = lc_last_error();
            lc_set_error("%s: %s", r->paths[(size_t)i].c_str(), why.c_str());
            return rc1;
        }
        if (info.num_frames && expect_dim > 0 && info.dim != expect_dim) {
            lc_set_error("%s: feature dim %d, expected %d", r->paths[(size_t)i].c_str(), info.dim, expect_dim);
            return LC_EINVAL;
        }
        num_frames[i] = info.num_frames;
        num_labels[i] = info.has_target ? info.num_labels : -1;       // -1: the record has no nnet_target list at all
        return LC_OK;
    }, err);
    if (rc != LC_OK) {
        lc_set_error("%s", err.c_str());
        delete r;
        return rc;
    }
    *reader = r;
    return LC_OK;
}

extern "C" int lc_batch_decode(lc_batch_reader_t *r, int left_context, int right_context, int subsample, float *x,
                               size_t utt_stride, size_t row_stride, int64_t max_rows, int64_t *labels,
                               size_t label_stride, int64_t max_labels, int64_t pad_label, int nthreads)
{
    if (!r || !x || r->dim <= 0) { lc_set_error("lc_batch_decode: bad argument"); return LC_EINVAL; }
    const int n = (int)r->raw.size();
    const size_t width = (size_t)r->dim * (size_t)(left_context + right_context + 1);
    std::string err;
    const int rc = run_parallel(n, nthreads, [&](int i) -> int {
        const auto &raw = r->raw[(size_t)i];
        float *xb = x + (size_t)i * utt_stride;
        int64_t *yb = labels ? labels + (size_t)i * label_stride : nullptr;
        lc_seqex_info_t info;
        int rc1 = lc_tfrecord_inspect(raw.data(), raw.size(), 0, &info);
        if (rc1 == LC_OK)
            rc1 = lc_tfrecord_decode(raw.data(), raw.size(), r->dim, left_context, right_context, subsample, xb, row_stride,
                                     max_rows, yb, max_labels);
        if (rc1 != LC_OK) {
            const std::string why = lc_last_error();
            lc_set_error("%s: %s", r->paths[(size_t)i].c_str(), why.c_str());
            return rc1;
        }
        const int64_t T = subsample > 0 ? info.num_frames / subsample : info.num_frames;
        for (int64_t j = T; j < max_rows; ++j) memset(xb + (size_t)j * row_stride, 0, width * sizeof(float));   // padding 0
        if (yb)
            for (int64_t j = info.has_target ? info.num_labels : 0; j < max_labels; ++j) yb[j] = pad_label;    // padding -1
        return LC_OK;
    }, err);
    if (rc != LC_OK) lc_set_error("%s", err.c_str());
    return rc;
}

extern "C" void lc_batch_close(lc_batch_reader_t *r) { delete r; }
