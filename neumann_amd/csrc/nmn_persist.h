// nmn_persist.h — on-disk sections shared by nmn_persist.hip (flat shard), nmn_ivf.hip (IVF) and nmn_engine.cpp
// (a collection's keys + metadata around a shard section).  Internal: never installed.
#pragma once
#include <cstdio>

#include "nmn_index.h"

struct nmn_ivf;

namespace nmn {

constexpr uint32_t kPersistFlat = 1;    // rows x dim f32 | rows f32 magnitudes
constexpr uint32_t kPersistIvf = 2;     // centroids | assign[] | a flat section with the vectors in id order
constexpr uint32_t kPersistEngine = 3;  // a collection of the host-side engine: config, keys, metadata | flat sections by dim

struct PersistHeader {  // 64 bytes, little endian
    char magic[8];          // "NMNIDX\0\1"
    uint32_t version;       // 1
    uint32_t kind;          // kPersist*
    uint32_t dim;
    uint32_t flags;
    uint64_t rows;          // entries of the section (flat: rows; ivf: vectors; engine: keys)
    uint64_t row_base;
    uint64_t payload_bytes; // bytes that follow this header and belong to the section (0 = not recorded)
    uint64_t aux;           // ivf: number of clusters; engine: bytes of the key / metadata block
    uint64_t reserved;
};
static_assert(sizeof(PersistHeader) == 64, "PersistHeader is part of the file format");

nmn_status persist_io_error(const char* what, const char* path);
// bytes between the file position and the end of the file (UINT64_MAX if it cannot be told): what a header may announce
uint64_t persist_bytes_left(FILE* fp);
nmn_status persist_check_file_size(const char* path, uint64_t max_file_bytes, uint64_t* size_out);
nmn_status persist_check_entries(uint64_t entries, uint64_t max_entries);
nmn_status persist_write_shard(nmn_index* idx, FILE* fp, const char* path);
nmn_status persist_read_header(FILE* fp, const char* path, PersistHeader* h);
nmn_status persist_read_rows_into(FILE* fp, const PersistHeader& h, nmn_index* idx);
nmn_status persist_read_shard(FILE* fp, const char* path, const PersistHeader& h, const nmn_index_desc* overrides,
                              nmn_index** out);

nmn_status persist_write_rows_host(FILE* fp, const char* path, uint32_t dim, uint64_t rows, uint64_t row_base,
                                   const float* tight_rows, const float* norms);
nmn_status persist_read_rows_host(FILE* fp, const PersistHeader& h, std::vector<float>* rows, std::vector<float>* norms);
nmn_status persist_write_ivf(nmn_ivf* ivf, FILE* fp, const char* path);
nmn_status persist_read_ivf(FILE* fp, const char* path, const PersistHeader& h, const nmn_index_desc* overrides, nmn_ivf** out);

}  // namespace nmn
