// nmn_persist.hip — persistence of the DEVICE LAYOUT of a shard (second half).
//
// The reference persists a collection as PersistentVectorIndex { collection, config, vectors: [(key, vector, metadata)],
// created_at, version } in JSON or bitcode (vector_engine/src/lib.rs:509-531, save/load 3794-3899) and guards a load
// with VectorEngineConfig::max_index_file_bytes / max_index_entries (lib.rs:644-646, 660-661: 100 MB / 1M entries by
// default; checked at 3831-3840 and 3847-3856).  What the GPU path adds to that is the matrix itself: the rows of a
// shard exactly as they sit in HBM (row-major f32, the stride removed) plus their reference-order magnitudes, so that a
// restart fills a shard with sequential reads and bulk H2D copies instead of per-key `store_embedding` calls.  Keys,
// metadata and collection config stay in the engine's file (nmn_engine.cpp: save_index_binary wraps this section).
//
// File = Header (64 bytes, little endian) | rows x dim f32, tightly packed | rows f32 magnitudes.
// The bf16 mirror is not stored: it is a pure function of the rows (8.6 ms for 10M x 768) and is rebuilt on the first
// search.  The magnitudes ARE stored and serve as the integrity check of a load: the upload recomputes them on the GPU in
// the reference's order (simd::magnitude, tensor_store/src/hnsw.rs:198-229) and every one must equal the file's bit for bit.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nmn_index.h"
#include "nmn_persist.h"

using namespace nmn;

#define P_TRY(expr)                                                   \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) return set_error_hip(_e, #expr);        \
    } while (0)

namespace nmn {

static const char kMagic[8] = {'N', 'M', 'N', 'I', 'D', 'X', 0, 1};

// 64-bit running checksum of a section's payload (rows, then magnitudes), kept in PersistHeader::reserved.  Four
// independent multiply-xor lanes over 8-byte words (ILP: ~10 GB/s on one core); order-sensitive, any single-bit change
// of the payload changes it.  The magnitude comparison after a load checks the DEVICE LAYOUT (what the GPU computed from
// the uploaded rows); this checks the bytes themselves — a flipped low mantissa bit moves no magnitude.
struct PersistHash {
    uint64_t lane[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    uint64_t count = 0;
    uint8_t carry[32];
    size_t n_carry = 0;
    void block(const uint8_t* p) {
        uint64_t w[4];
        memcpy(w, p, 32);
        for (int l = 0; l < 4; l++) {
            lane[l] = (lane[l] ^ w[l]) * 0x9E3779B97F4A7C15ull;
            lane[l] ^= lane[l] >> 29;
        }
    }
    void update(const void* data, size_t bytes) {  // the digest does not depend on how the payload is cut into calls
        const uint8_t* p = static_cast<const uint8_t*>(data);
        count += bytes;
        if (n_carry) {
            const size_t take = std::min(bytes, 32 - n_carry);
            memcpy(carry + n_carry, p, take);
            n_carry += take;
            p += take;
            bytes -= take;
            if (n_carry < 32) return;
            block(carry);
            n_carry = 0;
        }
        for (; bytes >= 32; p += 32, bytes -= 32) block(p);
        if (bytes) {
            memcpy(carry, p, bytes);
            n_carry = bytes;
        }
    }
    uint64_t digest() const {
        uint64_t h = count * 0xD6E8FEB86659FD93ull;
        uint64_t l2[4] = {lane[0], lane[1], lane[2], lane[3]};
        for (size_t i = 0; i < n_carry; i++) {
            l2[i & 3] = (l2[i & 3] ^ carry[i]) * 0x9E3779B97F4A7C15ull;
            l2[i & 3] ^= l2[i & 3] >> 29;
        }
        for (int l = 0; l < 4; l++) {
            h = (h ^ l2[l]) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 32;
        }
        return h ? h : 1;  // 0 = "no checksum recorded"
    }
};

nmn_status persist_io_error(const char* what, const char* path) {
    std::string m = std::string(what) + " '" + (path ? path : "") + "': " + strerror(errno);
    return set_error(NMN_ERR_IO, m.c_str());
}

// `max_index_file_bytes` (lib.rs:3831-3840): checked on the file's size BEFORE anything is read
nmn_status persist_check_file_size(const char* path, uint64_t max_file_bytes, uint64_t* size_out) {
    struct stat st;
    if (stat(path, &st) != 0) return persist_io_error("cannot stat", path);
    if (size_out) *size_out = (uint64_t)st.st_size;
    if (max_file_bytes && (uint64_t)st.st_size > max_file_bytes) {
        std::string m = "index file size " + std::to_string((uint64_t)st.st_size) + " exceeds limit " + std::to_string(max_file_bytes);
        return set_error(NMN_ERR_CONFIGURATION, m.c_str());
    }
    return NMN_OK;
}
// `max_index_entries` (lib.rs:3847-3856)
nmn_status persist_check_entries(uint64_t entries, uint64_t max_entries) {
    if (max_entries && entries > max_entries) {
        std::string m = "index entry count " + std::to_string(entries) + " exceeds limit " + std::to_string(max_entries);
        return set_error(NMN_ERR_CONFIGURATION, m.c_str());
    }
    return NMN_OK;
}

// rows [0, rows) of the shard and their magnitudes -> fp (current position).  Caller holds no lock.
nmn_status persist_write_shard(nmn_index* idx, FILE* fp, const char* path) {
    P_TRY(hipSetDevice(idx->device));
    // a search enqueued on another stream may still be converting rows; uploads are excluded by the caller's contract
    std::unique_lock<std::mutex> lk(idx->mu);
    PersistHeader h{};
    memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.kind = kPersistFlat;
    h.dim = idx->dim;
    h.flags = 0;
    h.rows = idx->rows;
    h.row_base = idx->row_base;
    h.payload_bytes = idx->rows * (uint64_t)idx->dim * 4ull + idx->rows * 4ull;
    const long header_pos = ftell(fp);
    if (header_pos < 0 || fwrite(&h, sizeof h, 1, fp) != 1) return persist_io_error("cannot write", path);
    PersistHash hash;
    hipStream_t s = idx->host_stream;
    P_TRY(hipStreamSynchronize(s));
    const size_t row_bytes = (size_t)idx->dim * 4;
    const uint64_t chunk_rows = std::max<uint64_t>(1, (32ull << 20) / row_bytes);  // 32 MiB of pinned staging
    uint8_t* pin = nullptr;
    P_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin), (size_t)chunk_rows * row_bytes, hipHostMallocDefault));
    nmn_status st = NMN_OK;
    for (uint64_t r = 0; r < idx->rows && st == NMN_OK; r += chunk_rows) {
        const uint64_t n = std::min(chunk_rows, idx->rows - r);
        hipError_t e = hipMemcpy2DAsync(pin, row_bytes, idx->corpus + r * (uint64_t)idx->ld, (size_t)idx->ld * 4, row_bytes, n,
                                        hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) st = set_error_hip(e, "reading the shard back");
        else if (fwrite(pin, row_bytes, n, fp) != n) st = persist_io_error("cannot write", path);
        else hash.update(pin, row_bytes * n);
    }
    for (uint64_t r = 0; r < idx->rows && st == NMN_OK; r += chunk_rows * idx->dim) {
        const uint64_t n = std::min<uint64_t>(chunk_rows * idx->dim, idx->rows - r);
        hipError_t e = hipMemcpyAsync(pin, idx->norms + r, n * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) st = set_error_hip(e, "reading the magnitudes back");
        else if (fwrite(pin, 4, n, fp) != n) st = persist_io_error("cannot write", path);
        else hash.update(pin, 4 * n);
    }
    (void)hipHostFree(pin);
    if (st == NMN_OK) {  // the checksum goes into the header, which was written first
        h.reserved = hash.digest();
        const long end_pos = ftell(fp);
        if (end_pos < 0 || fseek(fp, header_pos, SEEK_SET) != 0 || fwrite(&h, sizeof h, 1, fp) != 1 || fseek(fp, end_pos, SEEK_SET) != 0)
            st = persist_io_error("cannot write", path);
    }
    return st;
}

uint64_t persist_bytes_left(FILE* fp) {
    const long here = ftell(fp);
    if (here < 0 || fseek(fp, 0, SEEK_END) != 0) return UINT64_MAX;
    const long end = ftell(fp);
    if (fseek(fp, here, SEEK_SET) != 0 || end < here) return UINT64_MAX;
    return (uint64_t)(end - here);
}

nmn_status persist_read_header(FILE* fp, const char* path, PersistHeader* h) {
    if (fread(h, sizeof *h, 1, fp) != 1) return set_error(NMN_ERR_SERIALIZATION, "index file truncated (header)");
    if (memcmp(h->magic, kMagic, 8) != 0) return set_error(NMN_ERR_SERIALIZATION, "not a neumann_gpu index file (bad magic)");
    if (h->version != 1) return set_error(NMN_ERR_SERIALIZATION, "unsupported index file version");
    (void)path;
    // a header is only believed as far as the file can back it: the payload it announces must exist (a hostile header
    // must not size an allocation), and rows x dim must not wrap
    if (h->dim != 0 && h->rows > (UINT64_MAX / 8ull) / h->dim) return set_error(NMN_ERR_SERIALIZATION, "index file header is inconsistent");
    const long here = ftell(fp);
    if (here >= 0 && fseek(fp, 0, SEEK_END) == 0) {
        const long end = ftell(fp);
        if (fseek(fp, here, SEEK_SET) != 0) return set_error(NMN_ERR_IO, "IO error: cannot seek in the index file");
        if (end >= here && h->payload_bytes > (uint64_t)(end - here))
            return set_error(NMN_ERR_SERIALIZATION, "index file truncated (payload shorter than its header says)");
    }
    return NMN_OK;
}

// rows + magnitudes of a flat section (header already read into h) -> rows [0, h.rows) of an EXISTING shard of the same
// dimension; verifies the payload checksum and that the magnitudes the GPU computed equal the stored ones bit for bit
nmn_status persist_read_rows_into(FILE* fp, const PersistHeader& h, nmn_index* idx) {
    nmn_status st = NMN_OK;
    const size_t row_bytes = (size_t)h.dim * 4;
    const uint64_t chunk_rows = std::max<uint64_t>(1, (32ull << 20) / row_bytes);
    std::vector<float> buf((size_t)std::min<uint64_t>(chunk_rows, std::max<uint64_t>(h.rows, 1)) * h.dim);
    PersistHash hash;
    for (uint64_t r = 0; r < h.rows && st == NMN_OK; r += chunk_rows) {
        const uint64_t n = std::min(chunk_rows, h.rows - r);
        if (fread(buf.data(), row_bytes, n, fp) != n) st = set_error(NMN_ERR_SERIALIZATION, "index file truncated (rows)");
        else {
            hash.update(buf.data(), row_bytes * n);
            st = nmn_index_upload(idx, buf.data(), r, n);  // H2D + magnitudes in reference order
        }
    }
    // integrity: the magnitudes the GPU just computed must equal the stored ones bit for bit
    if (st == NMN_OK && h.rows) {
        std::vector<float> want((size_t)h.rows), got((size_t)h.rows);
        if (fread(want.data(), 4, h.rows, fp) != h.rows) st = set_error(NMN_ERR_SERIALIZATION, "index file truncated (magnitudes)");
        if (st == NMN_OK) {
            hash.update(want.data(), (size_t)h.rows * 4);
            if (h.reserved && hash.digest() != h.reserved) st = set_error(NMN_ERR_SERIALIZATION, "index file corrupt: checksum mismatch");
        }
        if (st == NMN_OK) {
            hipError_t e = hipSetDevice(idx->device);
            if (e == hipSuccess) e = hipMemcpy(got.data(), idx->norms, (size_t)h.rows * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) st = set_error_hip(e, "reading the magnitudes back");
            else if (memcmp(want.data(), got.data(), (size_t)h.rows * 4) != 0)
                st = set_error(NMN_ERR_SERIALIZATION, "index file corrupt: row magnitudes differ from the stored ones");
        }
    }
    return st;
}

// the section persist_write_shard wrote (header already read into h) -> a new shard
nmn_status persist_read_shard(FILE* fp, const char* path, const PersistHeader& h, const nmn_index_desc* over, nmn_index** out) {
    *out = nullptr;
    if (h.kind != kPersistFlat) return set_error(NMN_ERR_SERIALIZATION, "index file section is not a flat shard");
    if (h.dim == 0 || h.payload_bytes != h.rows * (uint64_t)h.dim * 4ull + h.rows * 4ull)
        return set_error(NMN_ERR_SERIALIZATION, "index file header is inconsistent");
    if (over && over->dim && over->dim != h.dim) {
        std::string m = "Dimension mismatch: expected " + std::to_string(over->dim) + ", got " + std::to_string(h.dim);
        return set_error(NMN_ERR_DIMENSION_MISMATCH, m.c_str());
    }
    nmn_index_desc d{};
    d.dim = h.dim;
    d.flags = over ? over->flags : 0;
    d.capacity_rows = std::max<uint64_t>(over ? over->capacity_rows : 0, std::max<uint64_t>(h.rows, 1));
    d.row_base = (over && over->row_base) ? over->row_base : h.row_base;
    d.device = over ? over->device : -1;
    d.cand_cap = over ? over->cand_cap : 0;
    nmn_index* idx = nullptr;
    nmn_status st = nmn_index_create(&d, &idx);
    if (st != NMN_OK) return st;
    st = persist_read_rows_into(fp, h, idx);
    if (st != NMN_OK) {
        const std::string keep = nmn_last_error();
        nmn_index_destroy(idx);
        set_error(st, keep.c_str());
        return st;
    }
    (void)path;
    *out = idx;
    return NMN_OK;
}

// a flat section from HOST rows (tightly packed) and their magnitudes: what the engine writes for a collection
nmn_status persist_write_rows_host(FILE* fp, const char* path, uint32_t dim, uint64_t rows, uint64_t row_base,
                                   const float* tight_rows, const float* norms) {
    PersistHeader h{};
    memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.kind = kPersistFlat;
    h.dim = dim;
    h.rows = rows;
    h.row_base = row_base;
    h.payload_bytes = rows * (uint64_t)dim * 4ull + rows * 4ull;
    PersistHash hash;
    hash.update(tight_rows, (size_t)rows * dim * 4);
    hash.update(norms, (size_t)rows * 4);
    h.reserved = hash.digest();
    if (fwrite(&h, sizeof h, 1, fp) != 1 || (rows && fwrite(tight_rows, (size_t)dim * 4, rows, fp) != rows) ||
        (rows && fwrite(norms, 4, rows, fp) != rows))
        return persist_io_error("cannot write", path);
    return NMN_OK;
}
nmn_status persist_read_rows_host(FILE* fp, const PersistHeader& h, std::vector<float>* rows, std::vector<float>* norms) {
    if (h.kind != kPersistFlat || h.dim == 0 || h.payload_bytes != h.rows * (uint64_t)h.dim * 4ull + h.rows * 4ull)
        return set_error(NMN_ERR_SERIALIZATION, "index file header is inconsistent");
    rows->resize((size_t)h.rows * h.dim);
    norms->resize((size_t)h.rows);
    if (h.rows && (fread(rows->data(), (size_t)h.dim * 4, h.rows, fp) != h.rows || fread(norms->data(), 4, h.rows, fp) != h.rows))
        return set_error(NMN_ERR_SERIALIZATION, "index file truncated (rows)");
    PersistHash hash;
    hash.update(rows->data(), rows->size() * 4);
    hash.update(norms->data(), norms->size() * 4);
    if (h.reserved && hash.digest() != h.reserved) return set_error(NMN_ERR_SERIALIZATION, "index file corrupt: checksum mismatch");
    return NMN_OK;
}

}  // namespace nmn

extern "C" nmn_status nmn_index_save(nmn_index* idx, const char* path) {
    if (!idx || !path) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    FILE* fp = fopen(path, "wb");
    if (!fp) return persist_io_error("cannot create", path);
    nmn_status st = persist_write_shard(idx, fp, path);
    if (fclose(fp) != 0 && st == NMN_OK) st = persist_io_error("cannot close", path);
    return st;
}

extern "C" nmn_status nmn_index_load(const char* path, const nmn_index_desc* overrides, uint64_t max_file_bytes,
                                     uint64_t max_entries, nmn_index** out) {
    if (!path || !out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_status st = persist_check_file_size(path, max_file_bytes, nullptr);
    if (st != NMN_OK) return st;
    FILE* fp = fopen(path, "rb");
    if (!fp) return persist_io_error("cannot open", path);
    PersistHeader h{};
    st = persist_read_header(fp, path, &h);
    if (st == NMN_OK) st = persist_check_entries(h.rows, max_entries);
    if (st == NMN_OK) st = persist_read_shard(fp, path, h, overrides, out);
    fclose(fp);
    return st;
}
