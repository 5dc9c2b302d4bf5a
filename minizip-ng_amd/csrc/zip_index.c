/* zip_index.c -- bulk central-directory indexer (host side, C).
 *
 * The batch path needs, for every entry of an archive, {method, flag, crc, sizes, payload offset} as one
 * flat table so that a whole archive becomes a single device launch.  The reference produces the same
 * facts one entry at a time (mz_zip_read_cd mz_zip.c:947-1100, mz_zip_entry_read_header :202-479,
 * mz_zip_goto_next_entry :2402-2412, local header skip :1874-1913) with ~25 small stream reads per record;
 * this is the "next" row f1 of SURVEY 8(f), restated as one pass over a memory image of the archive.
 * Format: doc/zip/appnote.txt sections 4.3.7 (local header), 4.3.12 (central header), 4.3.14-4.3.16
 * (ZIP64 end records / locator / end record), 4.5.3 (ZIP64 extended information extra field).
 * Row layout (8 x int64) equals integration/mz_driver.c drv_zip_index, which walks the archive with the
 * reference's own API -- tests/test_zip_index.py compares the two row for row.
 */
#include <string.h>

#include "mzhip.h"

static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

/* [off, off + len) lies inside a buffer of `total` bytes -- written so that nothing wraps */
static int fits(uint64_t off, uint64_t len, uint64_t total) { return off <= total && len <= total - off; }

#define SIG_LOCAL 0x04034b50u   /* mz_zip.c:59 */
#define SIG_CD 0x02014b50u      /* mz_zip.c:60 */
#define SIG_EOCD 0x06054b50u    /* mz_zip.c:61 */
#define SIG_EOCD64 0x06064b50u  /* mz_zip.c:62 */
#define SIG_LOC64 0x07064b50u   /* mz_zip.c:63 */

/* The walk itself, over bytes [base, zip_len) of the archive held at `buf` (buf[0] = byte `base` of the file): a whole image
 * (base = 0) or only its tail -- the end records and the central directory -- when the body is too large to hold
 * (shim_autoprime.c).  Local headers outside the bytes at hand leave a row's payload offset at -1 (mzhip_zip_index_resolve
 * fills it in from a window of the body).  *need_from: when the bytes at hand start too late (the end record names a
 * central directory in front of `base`, or no end record lies in them and the scan may go back further), the call returns
 * MZHIP_INDEX_NEED_MORE and the file offset from which the tail must be read. */
static int64_t index_core(const uint8_t *buf, uint64_t base, uint64_t zip_len, int64_t *table, int64_t max_entries, uint64_t *need_from) {
    if (!buf || zip_len < 22)
        return -103; /* MZ_FORMAT_ERROR */
    if (base > zip_len - 22) { /* not even the shortest end record is at hand */
        if (!need_from)
            return -103;
        *need_from = zip_len > 65536 + 22 ? zip_len - (65536 + 22) : 0;
        return MZHIP_INDEX_NEED_MORE;
    }
#define ZP(off) (buf + (size_t)((uint64_t)(off) - base)) /* byte `off` of the file; every use below is at an offset >= base */
#define HAVE(off, len) ((off) >= base && fits((off), (len), zip_len))
    /* end of central directory: scan back at most 1 MiB + comment (MZ_ZIP_EOCD_MAX_BACK, mz_zip.c:78-80) */
    uint64_t lo = zip_len > (1u << 20) + 22 ? zip_len - ((1u << 20) + 22) : 0;
    const uint64_t scan_lo = lo > base ? lo : base;
    int64_t eocd = -1;
    for (uint64_t i = zip_len - 22 + 1; i-- > scan_lo;) {
        if (rd32(ZP(i)) == SIG_EOCD) {
            eocd = (int64_t)i;
            break;
        }
    }
    if (eocd < 0) {
        if (base > lo && need_from) {
            *need_from = lo;
            return MZHIP_INDEX_NEED_MORE;
        }
        return -103;
    }
    uint64_t n_entries = rd16(ZP(eocd + 10));
    uint64_t cd_size = rd32(ZP(eocd + 12));
    uint64_t cd_off = rd32(ZP(eocd + 16));
    if (n_entries == 0xFFFF || cd_off == 0xFFFFFFFFu || cd_size == 0xFFFFFFFFu) {
        /* ZIP64: locator sits right before the EOCD (appnote 4.3.15) */
        if (eocd < 20)
            return -103;
        if (!HAVE((uint64_t)eocd - 20, 20)) {
            if (need_from) {
                *need_from = (uint64_t)eocd - 20;
                return MZHIP_INDEX_NEED_MORE;
            }
            return -103;
        }
        if (rd32(ZP(eocd - 20)) != SIG_LOC64)
            return -103;
        uint64_t e64 = rd64(ZP(eocd - 20 + 8));
        if (!fits(e64, 56, zip_len))
            return -103;
        if (e64 < base) {
            if (need_from) {
                *need_from = e64;
                return MZHIP_INDEX_NEED_MORE;
            }
            return -103;
        }
        if (rd32(ZP(e64)) != SIG_EOCD64)
            return -103;
        n_entries = rd64(ZP(e64 + 32));
        cd_size = rd64(ZP(e64 + 40));
        cd_off = rd64(ZP(e64 + 48));
    }
    if (!fits(cd_off, cd_size, zip_len))
        return -103;
    if (cd_off < base) {
        if (need_from) {
            *need_from = cd_off;
            return MZHIP_INDEX_NEED_MORE;
        }
        return -103;
    }
    uint64_t p = cd_off;
    int64_t n = 0;
    for (uint64_t k = 0; k < n_entries; k++) {
        if (!fits(p, 46, zip_len) || rd32(ZP(p)) != SIG_CD)
            return -103;
        const uint8_t *h = ZP(p);
        uint64_t flag = rd16(h + 8), method = rd16(h + 10), crc = rd32(h + 16);
        uint64_t csize = rd32(h + 20), usize = rd32(h + 24);
        uint32_t fn = rd16(h + 28), ex = rd16(h + 30), cm = rd16(h + 32);
        uint64_t disk = rd16(h + 34);
        uint64_t loff = rd32(h + 42);
        if (!fits(p, 46ull + fn + ex + cm, zip_len))
            return -103;
        /* ZIP64 extended information: only the fields that overflowed, in this fixed order (appnote 4.5.3) */
        const uint8_t *x = h + 46 + fn, *xe = x + ex;
        while (x + 4 <= xe) {
            uint32_t id = rd16(x), sz = rd16(x + 2);
            const uint8_t *q = x + 4;
            if (q + sz > xe)
                break;
            if (id == 0x0001) {
                if (usize == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { usize = rd64(q); q += 8; }
                if (csize == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { csize = rd64(q); q += 8; }
                if (loff == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { loff = rd64(q); q += 8; }
                if (disk == 0xFFFF && q + 4 <= x + 4 + sz) { disk = rd32(q); q += 4; }
            }
            x += 4 + sz;
        }
        /* a ZIP64 field that does not fit the reference's int64_t members is a format error there too
         * (mz_zip.c:328-339: `< 0` after the 64-bit read) */
        if ((usize | csize | loff) >> 63)
            return -103;
        int64_t payload = -1;
        if (HAVE(loff, 30) && rd32(ZP(loff)) == SIG_LOCAL) {
            uint64_t lfn = rd16(ZP(loff + 26)), lex = rd16(ZP(loff + 28));
            uint64_t pay = loff + 30 + lfn + lex; /* loff <= zip_len - 30: no wrap */
            if (fits(pay, csize, zip_len))
                payload = (int64_t)pay;
        }
        if (n < max_entries && table) {
            int64_t *t = table + n * 8;
            t[0] = (int64_t)method;
            t[1] = (int64_t)flag;
            t[2] = (int64_t)crc;
            t[3] = (int64_t)csize;
            t[4] = (int64_t)usize;
            t[5] = (int64_t)loff;
            t[6] = (int64_t)p; /* position of this record: what mz_zip_get_entry() returns (mz_zip.c:2368) */
            t[7] = payload;
        }
        n++;
        p += 46 + fn + ex + cm;
    }
#undef HAVE
#undef ZP
    return n;
}

int64_t mzhip_zip_index_mem(const uint8_t *zip, uint64_t zip_len, int64_t *table, int64_t max_entries) {
    return index_core(zip, 0, zip_len, table, max_entries, NULL);
}

/* The same from the archive's TAIL: tail[0] is byte tail_off of a file of zip_len bytes.  Rows whose local header lies in front
 * of the tail (all of them, normally) come back with payload offset -1.  MZHIP_INDEX_NEED_MORE: read the tail from *need_from. */
int64_t mzhip_zip_index_tail(const uint8_t *tail, uint64_t tail_off, uint64_t zip_len, int64_t *table, int64_t max_entries,
                             uint64_t *need_from) {
    if (tail_off > zip_len)
        return -102;
    return index_core(tail, tail_off, zip_len, table, max_entries, need_from);
}

/* Payload offsets from a WINDOW of the body: win[0] is byte win_off of the file, win_len bytes.  Every row of `table` (n rows of
 * 8) whose payload offset is unknown and whose local header AND payload lie inside the window gets it (local header skip as
 * mz_zip.c:1874-1913: 30 + name + extra of the LOCAL record).  Returns the number of rows resolved. */
int64_t mzhip_zip_index_resolve(const uint8_t *win, uint64_t win_off, uint64_t win_len, int64_t *table, int64_t n) {
    if (!win || !table)
        return -102;
    int64_t got = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t *t = table + i * 8;
        if (t[7] >= 0 || t[5] < 0 || t[3] < 0 || (uint64_t)t[5] < win_off)
            continue;
        const uint64_t rel = (uint64_t)t[5] - win_off;
        if (!fits(rel, 30, win_len) || rd32(win + rel) != SIG_LOCAL)
            continue;
        const uint64_t pay = rel + 30 + rd16(win + rel + 26) + rd16(win + rel + 28);
        if (!fits(pay, (uint64_t)t[3], win_len))
            continue;
        t[7] = (int64_t)(win_off + pay);
        got++;
    }
    return got;
}

/* The Hash extra field (0x1a51, mz.h:93; written by mz_zip_writer_entry_close, mz_zip_rw.c:1398-1408) of the entries of an
 * index: for row i of `table` (mzhip_zip_index_mem) the FIRST such field of its central-directory record -- what
 * mz_zip_reader_entry_get_first_hash picks (mz_zip_rw.c:510-540) -- as algorithm[i] (0 = the entry has none), digest_size[i]
 * and up to 64 digest bytes at digest + 64 * i.  Returns the number of entries that carry one, or < 0. */
static int64_t hash_core(const uint8_t *buf, uint64_t base, uint64_t zip_len, const int64_t *table, int64_t n, uint16_t *algorithm,
                         uint16_t *digest_size, uint8_t *digest) {
    if (!buf || !table || !algorithm || !digest_size || !digest || base > zip_len)
        return -102; /* MZ_PARAM_ERROR */
    int64_t found = 0;
    for (int64_t i = 0; i < n; i++) {
        algorithm[i] = 0;
        digest_size[i] = 0;
        memset(digest + 64 * i, 0, 64);
        const uint64_t p = (uint64_t)table[i * 8 + 6];
        if (p < base || !fits(p, 46, zip_len) || rd32(buf + (size_t)(p - base)) != SIG_CD)
            return -103;
        const uint8_t *h = buf + (size_t)(p - base);
        const uint32_t fn = rd16(h + 28), ex = rd16(h + 30);
        if (!fits(p, 46ull + fn + ex, zip_len))
            return -103;
        const uint8_t *x = h + 46 + fn, *xe = x + ex;
        while (x + 4 <= xe) {
            const uint32_t id = rd16(x), sz = rd16(x + 2);
            if (x + 4 + sz > xe)
                break;
            if (id == 0x1a51 && sz >= 4) {
                const uint32_t alg = rd16(x + 4), dsz = rd16(x + 6);
                algorithm[i] = (uint16_t)alg;
                digest_size[i] = (uint16_t)dsz;
                const uint32_t have = sz - 4 < dsz ? sz - 4 : dsz;
                memcpy(digest + 64 * i, x + 8, have < 64 ? have : 64);
                found++;
                break;
            }
            x += 4 + sz;
        }
    }
    return found;
}

int64_t mzhip_zip_index_hash_mem(const uint8_t *zip, uint64_t zip_len, const int64_t *table, int64_t n, uint16_t *algorithm,
                                 uint16_t *digest_size, uint8_t *digest) {
    return hash_core(zip, 0, zip_len, table, n, algorithm, digest_size, digest);
}

/* ... of a table made by mzhip_zip_index_tail: the central directory is in the tail */
int64_t mzhip_zip_index_hash_tail(const uint8_t *tail, uint64_t tail_off, uint64_t zip_len, const int64_t *table, int64_t n,
                                  uint16_t *algorithm, uint16_t *digest_size, uint8_t *digest) {
    return hash_core(tail, tail_off, zip_len, table, n, algorithm, digest_size, digest);
}

/* Contiguous slices of such a table for `world` devices ("shards naturally -- independent units"), balanced by
 * compressed + uncompressed bytes (+ 64 per entry): bounds[0 .. world], slice r = rows [bounds[r], bounds[r + 1]). */
void mzhip_shard_bounds(const int64_t *table, int64_t n, int32_t world, int64_t *bounds) {
    /* contiguous slices balanced by compressed + uncompressed bytes (+ 64 per entry): the rule of archive.shard_bounds */
    if (world < 1) world = 1;
    double total = 0;
    for (int64_t i = 0; i < n; i++) total += (double)(table[i * 8 + 3] + table[i * 8 + 4] + 64);
    bounds[0] = 0;
    double cum = 0;
    int64_t i = 0;
    for (int32_t r = 1; r < world; r++) {
        const double target = total * r / world;
        while (i < n && cum < target) {
            cum += (double)(table[i * 8 + 3] + table[i * 8 + 4] + 64);
            i++;
        }
        /* numpy.searchsorted(cum, target, 'left') over the cumulative sums that start with 0: first index with cum >= target */
        bounds[r] = i;
    }
    bounds[world] = n;
}
