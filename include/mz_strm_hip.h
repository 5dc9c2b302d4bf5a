/* mz_strm_hip.h -- the DROP-IN symbols of libmzhip.so.
 *
 * Every function below re-implements, with the same name, signature, return
 * convention and error behaviour, a symbol that minizip-ng's own objects
 * reference by name, so that the reference's mz_zip.c / mz_zip_rw.c /
 * mz_strm*.c / compat/ can be compiled UNMODIFIED and linked against
 * libmzhip.so in place of mz_strm_zlib.o, mz_strm_lzma.o and the CRC symbol
 * of mz_crypt.o (INTEGRATION.md has the link line):
 *
 *   replaced reference interface                      reference location
 *   -----------------------------------------------   -------------------------
 *   mz_stream_zlib_* (13 functions)                   mz_strm_zlib.h:20-35
 *   mz_stream_lzma_* (13 functions)                   mz_strm_lzma.h:20-35
 *   mz_crypt_crc32_update                             mz_crypt.h:20
 *   called from                                       mz_zip.c:1773,1792,2049,2064
 *   mz_crypt_sha_* (7 functions, optional)            mz_crypt.h:29-35
 *   called from                                       mz_zip_rw.c:409-451,462-467 (crypto builds)
 *
 * The stream-object contract they honour (mz_strm.h:53-72): the instance
 * starts with { vtbl*, base* }; vtbl has 12 slots in the order open, is_open,
 * read, write, tell, seek, close, error, create, destroy, get_prop_int64,
 * set_prop_int64; `base` is borrowed and is driven through its own vtbl.
 *
 * This header deliberately does not include any reference header; the few
 * constants it needs are restated with their source line.
 */
#ifndef MZ_STRM_HIP_H
#define MZ_STRM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MZHIP_API
#define MZHIP_API __attribute__((visibility("default")))
#endif

/* mz_strm.h:53-72 -- layout must match bit for bit */
typedef struct mzhip_stream_vtbl_s {
    int32_t (*open)(void *stream, const char *path, int32_t mode);
    int32_t (*is_open)(void *stream);
    int32_t (*read)(void *stream, void *buf, int32_t size);
    int32_t (*write)(void *stream, const void *buf, int32_t size);
    int64_t (*tell)(void *stream);
    int32_t (*seek)(void *stream, int64_t offset, int32_t origin);
    int32_t (*close)(void *stream);
    int32_t (*error)(void *stream);
    void *(*create)(void);
    void (*destroy)(void **stream);
    int32_t (*get_prop_int64)(void *stream, int32_t prop, int64_t *value);
    int32_t (*set_prop_int64)(void *stream, int32_t prop, int64_t value);
} mzhip_stream_vtbl;

typedef struct mzhip_stream_s {
    mzhip_stream_vtbl *vtbl;
    struct mzhip_stream_s *base;
} mzhip_stream;

/* mz_strm_zlib.h:20-35 */
MZHIP_API int32_t mz_stream_zlib_open(void *stream, const char *path, int32_t mode);
MZHIP_API int32_t mz_stream_zlib_is_open(void *stream);
MZHIP_API int32_t mz_stream_zlib_read(void *stream, void *buf, int32_t size);
MZHIP_API int32_t mz_stream_zlib_write(void *stream, const void *buf, int32_t size);
MZHIP_API int64_t mz_stream_zlib_tell(void *stream);
MZHIP_API int32_t mz_stream_zlib_seek(void *stream, int64_t offset, int32_t origin);
MZHIP_API int32_t mz_stream_zlib_close(void *stream);
MZHIP_API int32_t mz_stream_zlib_error(void *stream);
MZHIP_API int32_t mz_stream_zlib_get_prop_int64(void *stream, int32_t prop, int64_t *value);
MZHIP_API int32_t mz_stream_zlib_set_prop_int64(void *stream, int32_t prop, int64_t value);
MZHIP_API void *mz_stream_zlib_create(void);
MZHIP_API void mz_stream_zlib_delete(void **stream);
MZHIP_API void *mz_stream_zlib_get_interface(void);

/* mz_strm_lzma.h:20-35 */
MZHIP_API int32_t mz_stream_lzma_open(void *stream, const char *path, int32_t mode);
MZHIP_API int32_t mz_stream_lzma_is_open(void *stream);
MZHIP_API int32_t mz_stream_lzma_read(void *stream, void *buf, int32_t size);
MZHIP_API int32_t mz_stream_lzma_write(void *stream, const void *buf, int32_t size);
MZHIP_API int64_t mz_stream_lzma_tell(void *stream);
MZHIP_API int32_t mz_stream_lzma_seek(void *stream, int64_t offset, int32_t origin);
MZHIP_API int32_t mz_stream_lzma_close(void *stream);
MZHIP_API int32_t mz_stream_lzma_error(void *stream);
MZHIP_API int32_t mz_stream_lzma_get_prop_int64(void *stream, int32_t prop, int64_t *value);
MZHIP_API int32_t mz_stream_lzma_set_prop_int64(void *stream, int32_t prop, int64_t value);
MZHIP_API void *mz_stream_lzma_create(void);
MZHIP_API void mz_stream_lzma_delete(void **stream);
MZHIP_API void *mz_stream_lzma_get_interface(void);

/* mz_crypt.h:20 */
MZHIP_API uint32_t mz_crypt_crc32_update(uint32_t value, const uint8_t *buf, int32_t size);

/* mz_crypt.h:29-35 -- the hash the reader runs over an entry that carries a Hash extra field (mz_zip_rw.c:409-451,462-467).
 * For an entry that mzhip_prime_*() decoded, the digest is the one the device computed in that pass (shim_sha.c); for
 * everything else these call the reference's own implementation, which the link step keeps as mz_ref_crypt_sha_* (weak
 * here; absent: MZ_SUPPORT_ERROR -- this library holds no second SHA for a single stream).  Optional: an application
 * built with MZ_ZIP_NO_CRYPTO never references them. */
MZHIP_API void mz_crypt_sha_reset(void *handle);
MZHIP_API int32_t mz_crypt_sha_begin(void *handle);
MZHIP_API int32_t mz_crypt_sha_update(void *handle, const void *buf, int32_t size);
MZHIP_API int32_t mz_crypt_sha_end(void *handle, uint8_t *digest, int32_t digest_size);
MZHIP_API int32_t mz_crypt_sha_set_algorithm(void *handle, uint16_t algorithm);
MZHIP_API void *mz_crypt_sha_create(void);
MZHIP_API void mz_crypt_sha_delete(void **handle);

/* ---- constants restated from the reference (value, source) ---- */
#define MZH_OK 0               /* mz.h:21 */
#define MZH_STREAM_ERROR (-1)  /* mz.h:22 */
#define MZH_DATA_ERROR (-3)    /* mz.h:23 */
#define MZH_MEM_ERROR (-4)     /* mz.h:24 */
#define MZH_BUF_ERROR (-5)     /* mz.h:25 */
#define MZH_PARAM_ERROR (-102) /* mz.h:31 */
#define MZH_INTERNAL_ERROR (-104) /* mz.h:33 */
#define MZH_EXIST_ERROR (-107) /* mz.h:36 */
#define MZH_SUPPORT_ERROR (-109) /* mz.h:38 */
#define MZH_HASH_ERROR (-110)    /* mz.h:39 */
#define MZH_OPEN_ERROR (-111)  /* mz.h:40 */
#define MZH_CLOSE_ERROR (-112) /* mz.h:41 */
#define MZH_SEEK_ERROR (-113)  /* mz.h:42 */
#define MZH_TELL_ERROR (-114)  /* mz.h:43 */
#define MZH_WRITE_ERROR (-116) /* mz.h:45 */
#define MZH_OPEN_MODE_READ 0x01  /* mz.h:50 */
#define MZH_OPEN_MODE_WRITE 0x02 /* mz.h:51 */
#define MZH_COMPRESS_METHOD_LZMA 14 /* mz.h:66 */
#define MZH_COMPRESS_METHOD_XZ 95   /* mz.h:68 */
#define MZH_PROP_TOTAL_IN 1        /* mz_strm.h:20 */
#define MZH_PROP_TOTAL_IN_MAX 2    /* mz_strm.h:21 */
#define MZH_PROP_TOTAL_OUT 3       /* mz_strm.h:22 */
#define MZH_PROP_TOTAL_OUT_MAX 4   /* mz_strm.h:23 */
#define MZH_PROP_HEADER_SIZE 5     /* mz_strm.h:24 */
#define MZH_PROP_COMPRESS_LEVEL 9  /* mz_strm.h:28 */
#define MZH_PROP_COMPRESS_METHOD 10 /* mz_strm.h:29 */
#define MZH_PROP_COMPRESS_WINDOW 11 /* mz_strm.h:30 */
#define MZH_SEEK_SET 0 /* mz.h:58-60 */
#define MZH_SEEK_CUR 1
#define MZH_SEEK_END 2
#define MZH_STAGING_BYTES 32767    /* INT16_MAX staging reads, mz_strm_zlib.c:51,132 */

#ifdef __cplusplus
}
#endif
#endif
