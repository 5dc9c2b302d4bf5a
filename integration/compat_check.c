/* integration/compat_check.c -- the minizip 1.x compatibility API (compat/zip.h, compat/unzip.h) on top of the drop-in.
 *
 * Test infrastructure.  The reference's compat/zip.c, compat/unzip.c and compat/ioapi.c are compiled unmodified from
 * where they lie and linked with this file twice: against the HIP drop-in (integration/_build/compat_hip) and
 * against the reference codecs (oracle/_ref/compat_ref).  The checks restate test/test_compat.cc:23-53 (zip side)
 * and :55-139, :241-262 (unzip side); "big.bin" adds an entry large enough to cross several staging buffers.
 *
 *   compat_check write <zip>     create the archive through zipOpen64 / zipOpenNewFileInZip / zipWriteInFileInZip
 *   compat_check read  <zip>     walk it through unzOpen / unzLocateFile / unzReadCurrentFile / unztell / unzSeek64
 * Exit status = number of failed checks.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mz.h"
#include "mz_zip.h"
#include "unzip.h"
#include "zip.h"

static int failures = 0;
#define CHECK(cond, what)                                                      \
    do {                                                                       \
        if (!(cond)) {                                                         \
            failures++;                                                        \
            fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, what);      \
        }                                                                      \
    } while (0)

#define BIG_SIZE 300000u
static void fill_big(uint8_t *p) { /* word-like text from a small vocabulary: compresses about 3:1 */
    static const char *vocab[] = {"stream ", "entry ", "central ", "directory ", "offset ", "header ", "local ", "crc ", "\n"};
    uint32_t s = 12345u, n = 0;
    while (n < BIG_SIZE) {
        s = s * 1664525u + 1013904223u;
        const char *w = vocab[(s >> 24) % 9u];
        for (; *w && n < BIG_SIZE; w++) p[n++] = (uint8_t)*w;
    }
}

static void add_entry(zipFile zip, const char *name, const void *data, uint32_t size, uint32_t piece, int level) {
    zip_fileinfo fi;
    memset(&fi, 0, sizeof(fi));
    fi.mz_dos_date = mz_zip_time_t_to_dos_date(1588561637);
    int err = zipOpenNewFileInZip(zip, name, &fi, NULL, 0, NULL, 0, "test local comment", Z_DEFLATED, level);
    CHECK(err == ZIP_OK, "zipOpenNewFileInZip");
    if (err != ZIP_OK) return;
    for (uint32_t pos = 0; pos < size; pos += piece) {
        const uint32_t n = size - pos < piece ? size - pos : piece;
        CHECK(zipWriteInFileInZip(zip, (const uint8_t *)data + pos, n) == ZIP_OK, "zipWriteInFileInZip");
    }
    CHECK(zipCloseFileInZip(zip) == ZIP_OK, "zipCloseFileInZip");
}

static int do_write(const char *path) {
    uint8_t *big = (uint8_t *)malloc(BIG_SIZE);
    fill_big(big);
    zipFile zip = zipOpen64(path, APPEND_STATUS_CREATE);
    CHECK(zip != NULL, "zipOpen64");
    if (!zip) return failures;
    add_entry(zip, "test.txt", "test data", 9, 9, 1);
    add_entry(zip, "test2.txt", "test data", 9, 9, 0);
    add_entry(zip, "big.bin", big, BIG_SIZE, 7000, 6);
    CHECK(zipClose(zip, "test global comment") == ZIP_OK, "zipClose");
    free(big);
    return failures;
}

static int do_read(const char *path) {
    unz_global_info64 gi64;
    unz_global_info gi;
    unz_file_info64 fi64;
    unz_file_info fi;
    unz_file_pos fpos;
    char comment[120] = "", filename[120] = "", buffer[120];
    memset(&gi64, 0, sizeof(gi64));
    memset(&gi, 0, sizeof(gi));
    memset(&fi64, 0, sizeof(fi64));
    memset(&fi, 0, sizeof(fi));
    unzFile uz = unzOpen(path);
    CHECK(uz != NULL, "unzOpen");
    if (!uz) return failures;
    CHECK(unzGetGlobalComment(uz, comment, sizeof(comment)) == UNZ_OK, "unzGetGlobalComment");
    CHECK(strcmp(comment, "test global comment") == 0, "global comment text");
    CHECK(unzGetGlobalInfo(uz, &gi) == UNZ_OK && gi.number_entry == 3, "unzGetGlobalInfo");
    CHECK(unzGetGlobalInfo64(uz, &gi64) == UNZ_OK && gi64.number_entry == 3, "unzGetGlobalInfo64");
    CHECK(gi.number_disk_with_CD == 0 && gi64.number_disk_with_CD == 0, "disk with cd");
    CHECK(unzLocateFile(uz, "test.txt", 1) == UNZ_OK, "unzLocateFile");
    CHECK(unzGoToFirstFile(uz) == UNZ_OK, "unzGoToFirstFile");
    CHECK(unzGetCurrentFileInfo64(uz, &fi64, filename, sizeof(filename), NULL, 0, NULL, 0) == UNZ_OK, "info64");
    CHECK(strcmp(filename, "test.txt") == 0 && fi64.uncompressed_size == 9 && fi64.compression_method == Z_DEFLATED, "info64 fields");
    CHECK(unzOpenCurrentFile(uz) == UNZ_OK, "unzOpenCurrentFile");
    int got = unzReadCurrentFile(uz, buffer, sizeof(buffer));
    CHECK(got == 9 && memcmp(buffer, "test data", 9) == 0, "unzReadCurrentFile test.txt");
    CHECK(unzEndOfFile(uz) == 1, "unzEndOfFile");
    CHECK(unzCloseCurrentFile(uz) == UNZ_OK, "unzCloseCurrentFile (CRC verified here)");
    CHECK(unztell(uz) == got, "unztell");
    CHECK(unzGoToNextFile(uz) == UNZ_OK, "unzGoToNextFile");
    comment[0] = 0;
    CHECK(unzGetCurrentFileInfo(uz, &fi, filename, sizeof(filename), NULL, 0, comment, sizeof(comment)) == UNZ_OK, "info");
    CHECK(strcmp(comment, "test local comment") == 0 && strcmp(filename, "test2.txt") == 0, "entry comment");
    CHECK(fi.compression_method == 0, "level 0 is stored");
    CHECK(unzGetFilePos(uz, &fpos) == UNZ_OK && fpos.num_of_file == 1, "unzGetFilePos");
    CHECK(unzGetOffset(uz) > 0, "unzGetOffset");
    CHECK(unzOpenCurrentFile(uz) == UNZ_OK, "open stored");
    CHECK(unzReadCurrentFile(uz, buffer, sizeof(buffer)) == 9 && memcmp(buffer, "test data", 9) == 0, "read stored");
    CHECK(unzCloseCurrentFile(uz) == UNZ_OK, "close stored");
    /* the larger entry, in reads that do not divide anything */
    CHECK(unzGoToNextFile(uz) == UNZ_OK, "to big.bin");
    CHECK(unzGetCurrentFileInfo64(uz, &fi64, filename, sizeof(filename), NULL, 0, NULL, 0) == UNZ_OK, "big info");
    CHECK(strcmp(filename, "big.bin") == 0 && fi64.uncompressed_size == BIG_SIZE, "big fields");
    CHECK(fi64.compressed_size < BIG_SIZE / 2, "big.bin was compressed");
    {
        uint8_t *want = (uint8_t *)malloc(BIG_SIZE), *have = (uint8_t *)malloc(BIG_SIZE + 1000);
        uint32_t n = 0;
        fill_big(want);
        CHECK(unzOpenCurrentFile(uz) == UNZ_OK, "open big");
        for (;;) {
            int r = unzReadCurrentFile(uz, have + n, 1000);
            if (r <= 0) {
                CHECK(r == 0, "read big");
                break;
            }
            n += (uint32_t)r;
            if (n > BIG_SIZE) break;
        }
        CHECK(n == BIG_SIZE && memcmp(want, have, BIG_SIZE) == 0, "big.bin bytes");
        CHECK(unzeof(uz) == 1, "unzeof");
        CHECK(unztell64(uz) == BIG_SIZE, "unztell64");
        CHECK(unzCloseCurrentFile(uz) == UNZ_OK, "close big (CRC)");
        /* unzSeek64 works on stored entries only (compat/unzip.c) */
        free(want);
        free(have);
    }
    CHECK(unzGoToNextFile(uz) == UNZ_END_OF_LIST_OF_FILE, "end of list");
    CHECK(unzSeek64(uz, 0, SEEK_SET) == UNZ_PARAMERROR, "seek without an open entry");
    CHECK(unzClose(uz) == UNZ_OK, "unzClose");
    return failures;
}

int main(int argc, char **argv) {
    if (argc == 3 && strcmp(argv[1], "write") == 0) return do_write(argv[2]);
    if (argc == 3 && strcmp(argv[1], "read") == 0) return do_read(argv[2]);
    fprintf(stderr, "usage: %s write|read <zip>\n", argv[0]);
    return 99;
}
