/* nrvh_load_fast5: one single-read fast5 file -> the per-base arrays of one device call, in plain C.
 *
 * What it replaces, step for step (the Python host stage stays the definition and the fallback: whatever this file
 * does not recognise returns NRVH_UNSUPPORTED and the caller runs nanoreviser_amd/h5lite.py + hoststage.py, which give
 * the same numbers bit for bit - tests/test_hostlib_fast5.py):
 *   nanorev_fast5_handeler.py:58-83    open the file, `Analyses/<group>/<subgroup>/Events`, `Raw/Reads/<first>/Signal`,
 *                                      the `Fastq` record, the basecaller `version` attribute          (h5lite.read_fast5)
 *   nanorev_fast5_handeler.py:84-150   Events -> bases: move 0 dropped, move 2 two bases, lengths, rebasing
 *                                                                                                  (hoststage.collapse_events)
 *   preprocessing.py:100-101, 134-137  median / MAD of the samples, mean / std per base (NumPy's summation order)
 *                                                                                    (hoststage.median_mad, nrvh_event_stats)
 *   nanorevtrainutils.py:162-169       the six event features                                    (hoststage.feature_rows)
 * The HDF5 subset is the one h5lite.py implements from the published format specification: superblock 0/1, version-1
 * object headers with continuation blocks, symbol-table groups (v1 B-tree, SNOD, local heap), compound / fixed-point /
 * float / fixed-string datatypes, contiguous and chunked (v1 B-tree; deflate, shuffle) layouts.
 *
 * Why C: 1.9 ms per read and core in Python (inflate 0.5, event collapse 0.45, the HDF5 walk 0.45, copies 0.3) against
 * a GPU that revises a read in 0.6 ms - sixteen cores fed 56 M bases/s where eight GPUs take 85 M.  One
 * call per file, no Python object touched: ctypes releases the GIL, so the callers are THREADS of one process. */
#include <dlfcn.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/nanorev_host.h"

typedef struct {
  const uint8_t* d;
  size_t n;
  int bad;          /* set by any out-of-range access: the file is truncated / not what we parse */
  uint64_t work;    /* tree nodes / heap objects visited: a file whose links form a cycle runs out of budget, not of time */
} Buf;
#define WORK_BUDGET ((uint64_t)1 << 20)
static int spend(Buf* b) {
  if (++b->work > WORK_BUDGET) { b->bad = 1; return -1; }
  return 0;
}

static uint64_t U(Buf* b, uint64_t off, int nb) {
  if (off > b->n || (uint64_t)nb > b->n - off) { b->bad = 1; return 0; }
  uint64_t v = 0;
  for (int i = nb - 1; i >= 0; --i) v = (v << 8) | b->d[off + i];
  return v;
}
static const uint8_t* P(Buf* b, uint64_t off, uint64_t len) {
  if (off > b->n || len > b->n - off) { b->bad = 1; return 0; }
  return b->d + off;
}
#define UNDEF_ADDR 0xFFFFFFFFFFFFFFFFull
#define PAD8(x) (((x) + 7) & ~(uint64_t)7)

/* ---- object headers ------------------------------------------------------------------------------------------ */
typedef struct { int type; uint64_t off; uint64_t size; } Msg;
typedef struct { Msg m[96]; int n; } Obj;

static int read_obj(Buf* b, uint64_t addr, Obj* o) {
  o->n = 0;
  if (U(b, addr, 1) != 1) return -1;                    /* version-1 object header ("OHDR" = version 2: not ours) */
  const int nmsg = (int)U(b, addr + 2, 2);
  struct { uint64_t p, len; } blk[32];
  int nb = 0, ib = 0;
  blk[nb].p = addr + 16; blk[nb].len = U(b, addr + 8, 4); ++nb;
  while (ib < nb && o->n < nmsg && !b->bad) {
    uint64_t p = blk[ib].p, end = blk[ib].p + blk[ib].len;
    ++ib;
    while (p + 8 <= end && o->n < nmsg && !b->bad) {
      const int t = (int)U(b, p, 2);
      const uint64_t sz = U(b, p + 2, 2);
      if (t == 0x10 && nb < 32) { blk[nb].p = U(b, p + 8, 8); blk[nb].len = U(b, p + 16, 8); ++nb; }
      if (o->n >= 96) return -1;
      o->m[o->n].type = t; o->m[o->n].off = p + 8; o->m[o->n].size = sz; ++o->n;
      p += 8 + sz;
    }
  }
  return b->bad ? -1 : 0;
}
static const Msg* find_msg(const Obj* o, int type) {
  for (int i = 0; i < o->n; ++i) if (o->m[i].type == type) return &o->m[i];
  return 0;
}

/* ---- groups: symbol table -> v1 B-tree -> SNOD -> local heap -------------------------------------------------- */
/* name == NULL: the FIRST link in B-tree order is taken (h5lite: reads.keys()[0]) */
static int walk_group(Buf* b, uint64_t node, uint64_t hdata, const char* name, uint64_t* found, int depth) {
  if (depth > 16 || b->bad || spend(b)) return -1;
  const uint8_t* sig = P(b, node, 8);
  if (!sig) return -1;
  if (!memcmp(sig, "TREE", 4)) {
    const int used = (int)U(b, node + 6, 2);
    uint64_t p = node + 24;
    for (int i = 0; i < used; ++i, p += 16) {
      const int r = walk_group(b, U(b, p + 8, 8), hdata, name, found, depth + 1);
      if (r != 1) return r;                             /* 0 found, -1 error, 1 keep looking */
    }
    return 1;
  }
  if (!memcmp(sig, "SNOD", 4)) {
    const int nsym = (int)U(b, node + 6, 2);
    uint64_t p = node + 8;
    for (int i = 0; i < nsym; ++i, p += 40) {
      const uint64_t noff = U(b, p, 8), oaddr = U(b, p + 8, 8);
      if (!name) { *found = oaddr; return 0; }
      const size_t ln = strlen(name);
      const uint8_t* s = P(b, hdata + noff, ln + 1);
      if (s && !memcmp(s, name, ln) && s[ln] == 0) { *found = oaddr; return 0; }
    }
    return 1;
  }
  return -1;
}
static int group_lookup(Buf* b, uint64_t gaddr, const char* name, uint64_t* child) {
  Obj o;
  if (read_obj(b, gaddr, &o)) return -1;
  const Msg* st = find_msg(&o, 0x11);
  if (!st) return -1;
  const uint64_t btree = U(b, st->off, 8), heap = U(b, st->off + 8, 8);
  const uint8_t* hs = P(b, heap, 32);
  if (!hs || memcmp(hs, "HEAP", 4)) return -1;
  const uint64_t hdata = U(b, heap + 24, 8);
  const int r = walk_group(b, btree, hdata, name, child, 0);
  return r == 0 ? 0 : -1;
}
static int path_lookup(Buf* b, uint64_t root, const char* path, uint64_t* out) {
  char part[256];
  uint64_t cur = root;
  while (*path) {
    while (*path == '/') ++path;
    size_t n = 0;
    while (path[n] && path[n] != '/') ++n;
    if (!n) break;
    if (n >= sizeof part) return -1;
    memcpy(part, path, n); part[n] = 0;
    if (group_lookup(b, cur, part, &cur)) return -1;
    path += n;
  }
  *out = cur;
  return 0;
}

/* ---- datatypes ------------------------------------------------------------------------------------------------- */
typedef struct { int cls, size, is_signed; } Prim;     /* cls: 0 fixed point, 1 float, 3 fixed string */
typedef struct { char name[32]; int off; Prim t; } Member;

/* one datatype message at `off`: returns bytes consumed (>0) or -1.  Big-endian numbers -> -1 (never seen). */
static int64_t parse_type_d(Buf* b, uint64_t off, Prim* pt, Member* mem, int* nmem, int max_mem, int* total_size, int depth);
static int64_t parse_type(Buf* b, uint64_t off, Prim* pt, Member* mem, int* nmem, int max_mem, int* total_size) {
  return parse_type_d(b, off, pt, mem, nmem, max_mem, total_size, 0);
}
static int64_t parse_type_d(Buf* b, uint64_t off, Prim* pt, Member* mem, int* nmem, int max_mem, int* total_size, int depth) {
  if (depth > 4) return -1;                             /* vlen of vlen of ...: one level is all a fast5 has */
  const int cv = (int)U(b, off, 1), cls = cv & 15, ver = cv >> 4;
  const uint32_t bits = (uint32_t)U(b, off + 1, 3);
  const int size = (int)U(b, off + 4, 4);
  uint64_t p = off + 8;
  if (b->bad) return -1;
  if (total_size) *total_size = size;
  if (cls == 0) { if (bits & 1) return -1; pt->cls = 0; pt->size = size; pt->is_signed = (bits >> 3) & 1; return (int64_t)(p + 4 - off); }
  if (cls == 1) { if (bits & 1) return -1; pt->cls = 1; pt->size = size; pt->is_signed = 1; return (int64_t)(p + 12 - off); }
  if (cls == 3) { pt->cls = 3; pt->size = size; pt->is_signed = 0; return (int64_t)(p - off); }
  if (cls == 9) {                                       /* variable length: only strings (global heap), for attributes */
    if ((bits & 15) != 1) return -1;
    Prim base;
    const int64_t used = parse_type_d(b, p, &base, 0, 0, 0, 0, depth + 1);
    if (used < 0) return -1;
    pt->cls = 9; pt->size = size; pt->is_signed = 0;
    return (int64_t)(p + (uint64_t)used - off);
  }
  if (cls == 6 && mem) {
    const int nm = (int)(bits & 0xFFFF);
    *nmem = 0;
    for (int i = 0; i < nm; ++i) {
      const uint8_t* s = P(b, p, 1);
      if (!s) return -1;
      size_t ln = 0;
      while (p + ln < b->n && b->d[p + ln]) ++ln;
      if (p + ln >= b->n) return -1;
      Member m;
      memset(&m, 0, sizeof m);
      memcpy(m.name, b->d + p, ln < sizeof m.name - 1 ? ln : sizeof m.name - 1);
      p += ver < 3 ? PAD8(ln + 1) : ln + 1;
      if (ver == 1) { m.off = (int)U(b, p, 4); p += 4 + 1 + 3 + 4 + 4 + 16; }
      else if (ver == 2) { m.off = (int)U(b, p, 4); p += 4; }
      else {
        int nb = 1;
        while (nb < 4 && (size >> (8 * nb))) ++nb;
        m.off = (int)U(b, p, nb); p += nb;
      }
      Prim sub;
      const int64_t used = parse_type_d(b, p, &sub, 0, 0, 0, 0, depth + 1);
      if (used < 0) return -1;                          /* nested compound / vlen / ... : not ours */
      p += (uint64_t)used;
      m.t = sub;
      if (*nmem < max_mem) mem[(*nmem)++] = m;
    }
    pt->cls = 6; pt->size = size;
    return (int64_t)(p - off);
  }
  return -1;
}

/* ---- datasets -------------------------------------------------------------------------------------------------- */
typedef struct {
  Prim t; Member mem[24]; int nmem; int esize;
  int rank; uint64_t dims[4];
  int scalar;
  int layout;                 /* 1 contiguous, 2 chunked, 0 compact */
  uint64_t addr, size;        /* contiguous / compact data (compact: addr inside the header) */
  uint64_t btree; uint64_t cdim; /* chunked, rank 1: elements per chunk */
  int filt[4], nfilt; int shuffle_k;
} Dset;

static int read_dset(Buf* b, uint64_t addr, Dset* ds) {
  Obj o;
  memset(ds, 0, sizeof *ds);
  if (read_obj(b, addr, &o)) return -1;
  const Msg *sp = find_msg(&o, 0x01), *ty = find_msg(&o, 0x03), *la = find_msg(&o, 0x08), *fl = find_msg(&o, 0x0B);
  if (!sp || !ty || !la) return -1;
  { /* dataspace v1 / v2 */
    const int ver = (int)U(b, sp->off, 1), rank = (int)U(b, sp->off + 1, 1);
    uint64_t p;
    if (ver == 1) p = sp->off + 8;
    else if (ver == 2) { if (U(b, sp->off + 3, 1) == 2) return -1; p = sp->off + 4; }
    else return -1;
    if (rank > 4) return -1;
    ds->rank = rank; ds->scalar = rank == 0;
    for (int i = 0; i < rank; ++i) ds->dims[i] = U(b, p + 8 * i, 8);
  }
  if (parse_type(b, ty->off, &ds->t, ds->mem, &ds->nmem, 24, &ds->esize) < 0) return -1;
  if (ds->esize <= 0 || ds->esize > (1 << 20)) return -1;     /* an element of no bytes divides by zero further down */
  { /* layout */
    const uint64_t off = la->off;
    const int ver = (int)U(b, off, 1);
    if (ver == 3) {
      const int cls = (int)U(b, off + 1, 1);
      ds->layout = cls;
      if (cls == 0) { ds->size = U(b, off + 2, 2); ds->addr = off + 4; }
      else if (cls == 1) { ds->addr = U(b, off + 2, 8); ds->size = U(b, off + 10, 8); }
      else if (cls == 2) {
        const int rank = (int)U(b, off + 2, 1);
        if (rank != 2) return -1;                       /* 1-D data + the element-size dimension */
        ds->btree = U(b, off + 3, 8);
        ds->cdim = U(b, off + 11, 4);
      } else return -1;
    } else return -1;                                   /* layout v1 / v2 (pre-1.6 writers): the Python reader has them */
  }
  if (fl) {
    const uint64_t off = fl->off;
    const int ver = (int)U(b, off, 1), nf = (int)U(b, off + 1, 1);
    uint64_t p = off + (ver == 1 ? 8 : 2);
    if (nf > 4) return -1;
    for (int i = 0; i < nf; ++i) {
      const int fid = (int)U(b, p, 2);
      int ncv;
      if (ver == 1 || fid >= 256) {
        const uint64_t nlen = U(b, p + 2, 2);
        ncv = (int)U(b, p + 6, 2);
        p += 8 + (ver == 1 ? PAD8(nlen) : nlen);
      } else { ncv = (int)U(b, p + 4, 2); p += 6; }
      if (fid == 2) ds->shuffle_k = ncv > 0 ? (int)U(b, p, 4) : ds->esize;
      p += 4 * (uint64_t)ncv;
      if (ver == 1 && (ncv & 1)) p += 4;
      if (fid != 1 && fid != 2 && fid != 3) return -1;
      ds->filt[ds->nfilt++] = fid;
    }
  }
  return b->bad ? -1 : 0;
}

/* inflate: libdeflate when the image has it (about twice zlib's speed; the same bytes), zlib otherwise.  The callers
 * are threads: the library is looked up once (pthread_once), each thread owns one decompressor, freed when it ends. */
typedef void* (*ld_alloc_t)(void);
typedef void (*ld_free_t)(void*);
typedef int (*ld_dec_t)(void*, const void*, size_t, void*, size_t, size_t*);
static ld_alloc_t ld_alloc;
static ld_free_t ld_free;
static ld_dec_t ld_dec;
static int ld_usable;
static pthread_once_t ld_once = PTHREAD_ONCE_INIT;
static pthread_key_t ld_key;
static void ld_ctx_free(void* c) { if (c && ld_free) ld_free(c); }
static void ld_init(void) {
  void* h = getenv("NRV_NO_LIBDEFLATE") ? 0 : dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
  ld_alloc = h ? (ld_alloc_t)dlsym(h, "libdeflate_alloc_decompressor") : 0;
  ld_free = h ? (ld_free_t)dlsym(h, "libdeflate_free_decompressor") : 0;
  ld_dec = h ? (ld_dec_t)dlsym(h, "libdeflate_zlib_decompress") : 0;
  ld_usable = ld_alloc && ld_free && ld_dec && pthread_key_create(&ld_key, ld_ctx_free) == 0;
}
static int inflate_into(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* got) {
  pthread_once(&ld_once, ld_init);
  if (ld_usable) {
    void* ctx = pthread_getspecific(ld_key);
    if (!ctx) { ctx = ld_alloc(); if (ctx && pthread_setspecific(ld_key, ctx)) { ld_free(ctx); ctx = 0; } }
    if (ctx) {
      size_t out = 0;
      const int r = ld_dec(ctx, src, n, dst, cap, &out);         /* 0 = LIBDEFLATE_SUCCESS */
      if (r == 0) { *got = out; return 0; }
      /* short output buffer or anything else: let zlib have the last word */
    }
  }
  uLongf dl = (uLongf)cap;
  const int r = uncompress(dst, &dl, src, (uLong)n);
  if (r != Z_OK && r != Z_BUF_ERROR) return -1;
  if (r == Z_BUF_ERROR) return -1;                      /* a chunk larger than its declared size: not a valid file */
  *got = (size_t)dl;
  return 0;
}

/* the chunks of a 1-D chunked dataset into out[n_el * esize] */
static int walk_chunks(Buf* b, const Dset* ds, uint64_t node, uint8_t* out, uint8_t* tmp, uint8_t* tmp2, int depth) {
  if (node == UNDEF_ADDR) return 0;
  if (depth > 16 || spend(b)) return -1;
  const uint8_t* sig = P(b, node, 24);
  if (!sig || memcmp(sig, "TREE", 4) || U(b, node + 4, 1) != 1) return -1;
  const int level = (int)U(b, node + 5, 1), used = (int)U(b, node + 6, 2);
  const uint64_t ksz = 8 + 8 * 2;                       /* size, filter mask, offsets[rank + 1] with rank 1 */
  uint64_t p = node + 24;
  const size_t es = (size_t)ds->esize, need = (size_t)ds->cdim * es;
  for (int i = 0; i < used && !b->bad; ++i, p += ksz + 8) {
    const uint64_t csize = U(b, p, 4), fmask = U(b, p + 4, 4), off0 = U(b, p + 8, 8), child = U(b, p + ksz, 8);
    if (level > 0) { if (walk_chunks(b, ds, child, out, tmp, tmp2, depth + 1)) return -1; continue; }
    const uint8_t* data = P(b, child, csize);
    if (!data) return -1;
    size_t len = (size_t)csize;
    for (int f = ds->nfilt - 1; f >= 0; --f) {           /* the pipeline in reverse */
      if (fmask & (1u << f)) continue;
      if (ds->filt[f] == 1) {
        size_t got = 0;
        uint8_t* dst = data == tmp ? tmp2 : tmp;
        if (inflate_into(data, len, dst, need, &got)) return -1;
        data = dst; len = got;
      } else if (ds->filt[f] == 2) {
        const size_t k = (size_t)(ds->shuffle_k > 0 ? ds->shuffle_k : (int)es), m = len / k;
        uint8_t* dst = data == tmp ? tmp2 : tmp;
        if (k * m > need) return -1;
        for (size_t e = 0; e < m; ++e) for (size_t j = 0; j < k; ++j) dst[e * k + j] = data[j * m + e];
        data = dst; len = k * m;
      } else if (ds->filt[f] == 3) { if (len < 4) return -1; len -= 4; }
    }
    if (off0 >= ds->dims[0]) continue;
    uint64_t cnt = ds->cdim;
    if (off0 + cnt > ds->dims[0]) cnt = ds->dims[0] - off0;
    size_t have = len / es;                              /* writers may store a short edge chunk: the rest stays 0 */
    if (have > cnt) have = (size_t)cnt;
    memcpy(out + off0 * es, data, have * es);
  }
  return b->bad ? -1 : 0;
}

/* raw bytes of a dataset (caller frees *out when *owned) */
static int dset_bytes(Buf* b, const Dset* ds, const uint8_t** out, size_t* nbytes, int* owned) {
  uint64_t n = 1;
  if (ds->esize <= 0) return -1;
  for (int i = 0; i < ds->rank; ++i) {                  /* n x esize must not wrap: a file is never 2^40 bytes */
    if (ds->dims[i] != 0 && n > ((uint64_t)1 << 40) / ds->dims[i]) return -1;
    n *= ds->dims[i];
  }
  if (n > ((uint64_t)1 << 40) / (uint64_t)ds->esize) return -1;
  const uint64_t want = n * (uint64_t)ds->esize;
  *owned = 0;
  if (ds->layout == 0 || ds->layout == 1) {
    if (ds->layout == 1 && ds->addr == UNDEF_ADDR) return -1;
    const uint8_t* p = P(b, ds->addr, want);
    if (!p) return -1;
    *out = p; *nbytes = (size_t)want;
    return 0;
  }
  /* deflate expands at most ~1032 : 1, so a dataset cannot be larger than that many times the file; a chunk is never 2 GB */
  if (ds->rank != 1 || ds->cdim == 0 || want > ((uint64_t)1 << 34) || want / 1032 > (uint64_t)b->n + 4096) return -1;
  if ((uint64_t)ds->cdim * (uint64_t)ds->esize > ((uint64_t)1 << 31)) return -1;
  const size_t need = (size_t)ds->cdim * (size_t)ds->esize;
  uint8_t* buf = (uint8_t*)calloc((size_t)want + 1, 1);
  uint8_t* tmp = (uint8_t*)malloc(2 * need + 16);
  if (!buf || !tmp) { free(buf); free(tmp); return -1; }
  const int r = walk_chunks(b, ds, ds->btree, buf, tmp, tmp + need + 8, 0);
  free(tmp);
  if (r) { free(buf); return -1; }
  *out = buf; *nbytes = (size_t)want; *owned = 1;
  return 0;
}

/* the basecaller's `version` attribute: old (<= 0.0) files keep event times in seconds (nanorev_fast5_handeler.py:66-71):
 * 1 "new" (a dotted version above 0.0), 0 anything we would have to think about */
static int version_is_new(Buf* b, uint64_t gaddr) {
  Obj o;
  if (read_obj(b, gaddr, &o)) return 0;
  for (int i = 0; i < o.n; ++i) {
    if (o.m[i].type != 0x0C) continue;
    const uint64_t off = o.m[i].off;
    const int ver = (int)U(b, off, 1);
    const uint64_t nsz = U(b, off + 2, 2), tsz = U(b, off + 4, 2);
    uint64_t p = off + 8 + (ver == 3 ? 1 : 0);
    const uint8_t* nm = P(b, p, nsz);
    if (!nm || nsz < 8 || memcmp(nm, "version", 8)) continue;
    p += ver == 1 ? PAD8(nsz) : nsz;
    Prim t;
    if (parse_type(b, p, &t, 0, 0, 0, 0) < 0 || (t.cls != 3 && t.cls != 9)) return 0;
    const uint64_t ssz = U(b, off + 6, 2);
    p += ver == 1 ? PAD8(tsz) : tsz;
    p += ver == 1 ? PAD8(ssz) : ssz;
    const uint8_t* s = P(b, p, (uint64_t)t.size);
    if (!s) return 0;
    char v[32] = {0};
    if (t.cls == 3) memcpy(v, s, t.size < 31 ? (size_t)t.size : 31);
    else {                                              /* {length, global heap collection, object index} */
      if (t.size != 16) return 0;
      const uint64_t ln = U(b, p, 4), gcol = U(b, p + 4, 8), idx = U(b, p + 12, 4);
      const uint8_t* g = P(b, gcol, 16);
      if (!g || memcmp(g, "GCOL", 4)) return 0;
      const uint64_t gend = gcol + U(b, gcol + 8, 8);
      uint64_t q = gcol + 16;
      int hit = 0;
      while (q + 16 <= gend && !b->bad && !spend(b)) {
        const uint64_t oi = U(b, q, 2), osz = U(b, q + 8, 8);
        if (oi == 0) break;
        if (oi == idx) {
          const uint64_t take = ln < osz ? ln : osz;
          const uint8_t* sp = P(b, q + 16, take);
          if (!sp) return 0;
          memcpy(v, sp, take < 31 ? (size_t)take : 31);
          hit = 1;
          break;
        }
        q += 16 + PAD8(osz);
      }
      if (!hit) return 0;
    }
    int a = -1, c = -1;
    if (sscanf(v, "%d.%d", &a, &c) != 2) return 0;
    return (a > 0 || (a == 0 && c > 0)) ? 1 : 0;
  }
  return 0;                                             /* no attribute: h5lite defaults to "0.0" = old */
}

#define START_LIM ((int64_t)1 << 40)
static int int_size_ok(const Prim* t) { return t->size == 1 || t->size == 2 || t->size == 4 || t->size == 8; }
/* callers check int_size_ok first.  One branch per width with a FIXED-size memcpy (a move): with the width as a run-time
 * argument this was a library call per field, ~0.6 ms per read of 20 k events - the largest item of the host stage after the
 * inflate (r06 profile) */
static inline int64_t load_int(const uint8_t* p, const Prim* t) {
  switch (t->size) {
    case 4: { uint32_t v; memcpy(&v, p, 4); return t->is_signed ? (int64_t)(int32_t)v : (int64_t)v; }
    case 8: { uint64_t v; memcpy(&v, p, 8); return (int64_t)v; }
    case 2: { uint16_t v; memcpy(&v, p, 2); return t->is_signed ? (int64_t)(int16_t)v : (int64_t)v; }
    default: { const uint8_t v = *p; return t->is_signed ? (int64_t)(int8_t)v : (int64_t)v; }
  }
}

/* median and median absolute deviation of int16 samples through histograms (hoststage.median_mad: exact) */
static void two_middle(const uint32_t* cnt, int64_t nbin, int64_t n, int64_t* lo, int64_t* hi) {
  int64_t cum = 0, l = -1, h = -1;
  for (int64_t v = 0; v < nbin; ++v) {
    cum += cnt[v];
    if (l < 0 && cum >= n / 2) l = v;                   /* np.searchsorted(cum, n // 2): first index with cum >= n // 2 */
    if (cum >= n / 2 + 1) { h = v; break; }
  }
  *hi = h;
  *lo = (n & 1) ? h : l;
}
static int median_mad_i16(const int16_t* x, int64_t n, double* shift, double* scale) {
  if (n <= 0 || n >= ((int64_t)1 << 32)) return -1;
  int base = x[0], top = x[0];
  for (int64_t i = 1; i < n; ++i) { const int v = x[i]; base = v < base ? v : base; top = v > top ? v : top; }
  const int64_t range = (int64_t)top - base + 1, nb2 = 2 * range + 2;      /* |2 x - 2 shift| <= 2 (top - base) */
  /* FOUR histograms, sample i into histogram i & 3: neighbouring samples of a nanopore trace are often EQUAL, and a run of
   * increments of one counter waits for its own store each time (r06 profile: 0.50 -> 0.3 ms per read); summed before the scan */
  uint32_t* cnt = (uint32_t*)calloc((size_t)nb2 * 4, sizeof(uint32_t));
  if (!cnt) return -1;
  uint32_t *c0 = cnt, *c1 = cnt + nb2, *c2 = cnt + 2 * nb2, *c3 = cnt + 3 * nb2;
  int64_t i = 0;
  for (; i + 4 <= n; i += 4) { ++c0[x[i] - base]; ++c1[x[i + 1] - base]; ++c2[x[i + 2] - base]; ++c3[x[i + 3] - base]; }
  for (; i < n; ++i) ++c0[x[i] - base];
  for (int64_t v = 0; v < range; ++v) c0[v] += c1[v] + c2[v] + c3[v];
  int64_t lo, hi;
  two_middle(c0, range, n, &lo, &hi);
  const int64_t s2 = lo + hi + 2 * (int64_t)base;       /* 2 x shift */
  memset(cnt, 0, (size_t)nb2 * 4 * sizeof(uint32_t));
  const int s2i = (int)s2;                              /* |s2| <= 2^17: fits */
  for (i = 0; i + 4 <= n; i += 4) {
    const int k0 = 2 * x[i] - s2i, k1 = 2 * x[i + 1] - s2i, k2 = 2 * x[i + 2] - s2i, k3 = 2 * x[i + 3] - s2i;
    ++c0[k0 < 0 ? -k0 : k0]; ++c1[k1 < 0 ? -k1 : k1]; ++c2[k2 < 0 ? -k2 : k2]; ++c3[k3 < 0 ? -k3 : k3];
  }
  for (; i < n; ++i) { const int k = 2 * x[i] - s2i; ++c0[k < 0 ? -k : k]; }
  for (int64_t v = 0; v < nb2; ++v) c0[v] += c1[v] + c2[v] + c3[v];
  int64_t klo, khi;
  two_middle(c0, nb2, n, &klo, &khi);
  free(cnt);
  *shift = (double)s2 / 2.0;
  *scale = (double)(klo + khi) / 4.0;
  return 0;
}

void nrvh_free_read(nrvh_read* r) {
  if (!r) return;
  free(r->raw); free(r->starts); free(r->feat); free(r->bases); free(r->fastq);
  memset(r, 0, sizeof *r);
}

static int fail(char* err, int err_len, int code, const char* msg) {
  if (err && err_len > 0) { strncpy(err, msg, (size_t)err_len - 1); err[err_len - 1] = 0; }
  return code;
}

/* the image of one file (fsz bytes, not modified, not kept) -> *out; the sanitizer driver (tools/hostfuzz) enters here */
static int load_fast5_image(const uint8_t* file, long fsz, const char* group, const char* subgroup, int want_fastq,
                            nrvh_read* out, char* err, int err_len) {
  memset(out, 0, sizeof *out);
  Buf B = {file, (size_t)fsz, 0, 0};
  Buf* b = &B;
  int rc = NRVH_UNSUPPORTED;
  const char* why = "not the HDF5 subset of the native reader";
  const uint8_t *ev_bytes = 0, *sig_bytes = 0, *fq_bytes = 0;
  size_t ev_n = 0, sig_n = 0, fq_n = 0;
  int ev_owned = 0, sig_owned = 0, fq_owned = 0;
  int64_t* st64 = 0;
  double *mean = 0, *sd = 0, *len64 = 0;
  float *abm = 0, *abs_ = 0;

  do {
    static const uint8_t kSig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    if (fsz < 96 || memcmp(file, kSig, 8)) break;
    const int sver = file[8];
    if (sver > 1 || file[13] != 8 || file[14] != 8) break;
    const uint64_t root_entry = 24 + (sver == 1 ? 4 : 0) + 32;
    const uint64_t root = U(b, root_entry + 8, 8);
    char pth[600];
    uint64_t gaddr, eaddr, raddr, rd0, saddr, faddr;
    snprintf(pth, sizeof pth, "Analyses/%s", group);
    if (path_lookup(b, root, pth, &gaddr)) break;       /* missing group: the Python path raises the reference's message */
    if (!version_is_new(b, gaddr)) { why = "pre-versioned basecaller output (times in seconds)"; break; }
    snprintf(pth, sizeof pth, "%s/Events", subgroup);
    if (path_lookup(b, gaddr, pth, &eaddr)) break;
    if (path_lookup(b, root, "Raw/Reads", &raddr) || group_lookup(b, raddr, 0, &rd0) || group_lookup(b, rd0, "Signal", &saddr)) break;
    Dset ev, sg, fq;
    if (read_dset(b, eaddr, &ev) || ev.t.cls != 6 || ev.rank != 1) break;
    if (read_dset(b, saddr, &sg) || sg.t.cls != 0 || sg.esize != 2 || !sg.t.is_signed || sg.rank != 1) { why = "Signal is not a 1-D int16 dataset"; break; }
    const Member *m_start = 0, *m_mean = 0, *m_stdv = 0, *m_state = 0, *m_move = 0;
    for (int i = 0; i < ev.nmem; ++i) {
      const Member* m = &ev.mem[i];
      if (!strcmp(m->name, "start")) m_start = m;
      else if (!strcmp(m->name, "mean")) m_mean = m;
      else if (!strcmp(m->name, "stdv")) m_stdv = m;
      else if (!strcmp(m->name, "model_state")) m_state = m;
      else if (!strcmp(m->name, "move")) m_move = m;
    }
    if (!m_start || !m_mean || !m_stdv || !m_state || !m_move) break;
    if (m_start->t.cls != 0 || m_move->t.cls != 0 || m_state->t.cls != 3 || m_state->t.size < 3) { why = "Events fields of another type"; break; }
    if (m_mean->t.cls != 1 || m_stdv->t.cls != 1 || m_mean->t.size != 4 || m_stdv->t.size != 4) { why = "Events mean / stdv are not float32"; break; }
    /* sizes and offsets come from the FILE: every field that is read must lie inside a row, integers must fit load_int */
    if (!int_size_ok(&m_start->t) || !int_size_ok(&m_move->t)) { why = "Events integer fields of an unusual width"; break; }
    {
      const Member* used[5] = {m_start, m_mean, m_stdv, m_state, m_move};
      int inside = ev.esize > 0;
      for (int i = 0; i < 5 && inside; ++i)
        inside = used[i]->off >= 0 && used[i]->t.size > 0 && (int64_t)used[i]->off + used[i]->t.size <= (int64_t)ev.esize;
      if (!inside) { why = "Events field outside its row"; break; }
    }
    if (dset_bytes(b, &ev, &ev_bytes, &ev_n, &ev_owned)) break;
    if (dset_bytes(b, &sg, &sig_bytes, &sig_n, &sig_owned)) break;
    const int64_t n_ev_in = (int64_t)ev.dims[0], es = ev.esize, L = (int64_t)sg.dims[0];

    /* ---- Events -> bases (hoststage.collapse_events) */
    int64_t n = 0;
    for (int64_t i = 0; i < n_ev_in; ++i) {
      const int64_t mv = load_int(ev_bytes + i * es + m_move->off, &m_move->t);
      n += mv == 0 ? 0 : (mv == 2 ? 2 : 1);
    }
    rc = NRVH_E_READ; why = "Events is too short or there are too much zero moves.";
    if (n < 2) break;
    st64 = (int64_t*)malloc((size_t)n * 8); len64 = (double*)malloc((size_t)n * 8);
    abm = (float*)malloc((size_t)n * 4); abs_ = (float*)malloc((size_t)n * 4);
    out->bases = (char*)malloc((size_t)n + 1);
    if (!st64 || !len64 || !abm || !abs_ || !out->bases) { rc = NRVH_E_IO; why = "out of memory"; break; }
    int64_t k = 0;
    int wild = 0;                                       /* a start no DAQ produces: differences of those would overflow */
    for (int64_t i = 0; i < n_ev_in; ++i) {
      const uint8_t* row = ev_bytes + i * es;
      const int64_t mv = load_int(row + m_move->off, &m_move->t);
      if (mv == 0) continue;
      const int64_t s = load_int(row + m_start->off, &m_start->t);
      if (s < -START_LIM || s > START_LIM) { wild = 1; break; }
      float fm, fs;
      memcpy(&fm, row + m_mean->off, 4); memcpy(&fs, row + m_stdv->off, 4);
      const char* ms = (const char*)row + m_state->off;
      if (mv == 2) {
        st64[k] = s; out->bases[k] = ms[1]; abm[k] = fm; abs_[k] = fs; ++k;
        st64[k] = s + 2; out->bases[k] = ms[2]; abm[k] = fm; abs_[k] = fs; ++k;
      } else {
        st64[k] = s; out->bases[k] = ms[2]; abm[k] = fm; abs_[k] = fs; ++k;
      }
    }
    if (wild) { rc = NRVH_UNSUPPORTED; why = "event starts outside the native reader's range"; break; }
    out->bases[n] = 0;
    for (int64_t i = 0; i + 1 < n; ++i) len64[i] = (double)(st64[i + 1] - st64[i]);
    len64[n - 1] = st64[n - 1] - st64[n - 2] < 5 ? 3.0 : 5.0;
    why = "Signal is shorter than the Events";
    if (L < st64[n - 1] + (int64_t)len64[n - 1]) break;
    const int64_t a0 = st64[0];
    if (a0 < 0 || a0 > L || st64[n - 1] - a0 + 8 >= ((int64_t)1 << 31)) { rc = NRVH_UNSUPPORTED; why = "event starts outside the native reader's range"; break; }
    rc = NRVH_E_IO; why = "out of memory";
    /* ---- samples from the first event on; starts relative to it */
    out->n_raw = L - a0;
    out->n_ev = n;
    out->raw = (int16_t*)malloc((size_t)(out->n_raw > 0 ? out->n_raw : 1) * 2);
    out->starts = (int32_t*)malloc((size_t)n * 4);
    out->feat = (float*)malloc((size_t)n * 6 * 4);
    mean = (double*)malloc((size_t)n * 8); sd = (double*)malloc((size_t)n * 8);
    if (!out->raw || !out->starts || !out->feat || !mean || !sd) break;
    memcpy(out->raw, sig_bytes + a0 * 2, (size_t)out->n_raw * 2);
    int mono = 1;
    for (int64_t i = 0; i < n; ++i) {
      if (st64[i] < a0 || st64[i] - a0 >= ((int64_t)1 << 31)) { mono = 0; break; }
      out->starts[i] = (int32_t)(st64[i] - a0);
    }
    if (!mono) { rc = NRVH_UNSUPPORTED; why = "event starts not ascending"; break; }
    /* ---- shift / scale, per-base mean / std (preprocessing.py:100-101, 134-137) */
    if (median_mad_i16(out->raw, out->n_raw, &out->shift, &out->scale)) { rc = NRVH_UNSUPPORTED; why = "empty signal behind the first event"; break; }
    if (nrvh_event_stats(out->raw, out->n_raw, out->starts, n, (int32_t)len64[n - 1], mean, sd)) { rc = NRVH_UNSUPPORTED; why = "event statistics"; break; }
    /* ---- the six features, f64 arithmetic, stored as f32 (hoststage.feature_rows + astype(float32)) */
    for (int64_t i = 0; i < n; ++i) {
      double col = 0.0;
      switch (out->bases[i]) { case 'A': col = 250; break; case 'G': col = 180; break; case 'T': col = 100; break; case 'C': col = 30; break; default: break; }
      float* f = out->feat + i * 6;
      f[0] = (float)(col / 300.0);
      f[1] = (float)(mean[i] / out->shift);
      f[2] = (float)(sd[i] / out->scale);
      f[3] = (float)(len64[i] / 10.0);
      f[4] = (float)(double)abm[i];
      f[5] = (float)(double)abs_[i];
    }
    /* ---- the Fastq record (only when the caller writes FASTQ, or wants it for its failure path) */
    if (want_fastq) {
      snprintf(pth, sizeof pth, "%s/Fastq", subgroup);
      if (!path_lookup(b, gaddr, pth, &faddr)) {
        if (read_dset(b, faddr, &fq) || fq.t.cls != 3 || !fq.scalar) { rc = NRVH_UNSUPPORTED; why = "Fastq is not a fixed-length string"; break; }
        fq.esize = fq.t.size;
        if (dset_bytes(b, &fq, &fq_bytes, &fq_n, &fq_owned)) { rc = NRVH_UNSUPPORTED; why = "Fastq unreadable"; break; }
        while (fq_n > 0 && fq_bytes[fq_n - 1] == 0) --fq_n;          /* numpy's S dtype drops trailing NULs */
        out->fastq = (char*)malloc(fq_n + 1);
        if (!out->fastq) break;
        memcpy(out->fastq, fq_bytes, fq_n);
        out->fastq[fq_n] = 0;
        out->fastq_len = (int64_t)fq_n;
      }
    }
    rc = b->bad ? NRVH_UNSUPPORTED : NRVH_OK;
    why = "truncated file";
  } while (0);

  if (ev_owned) free((void*)ev_bytes);
  if (sig_owned) free((void*)sig_bytes);
  if (fq_owned) free((void*)fq_bytes);
  free(st64); free(len64); free(abm); free(abs_); free(mean); free(sd);
  if (rc != NRVH_OK) { nrvh_free_read(out); return fail(err, err_len, rc, why); }
  return NRVH_OK;
}

int nrvh_load_fast5(const char* path, const char* group, const char* subgroup, int want_fastq, nrvh_read* out,
                    char* err, int err_len) {
  if (!path || !group || !subgroup || !out) return fail(err, err_len, NRVH_E_ARG, "bad arguments");
  memset(out, 0, sizeof *out);
  /* The file is MAPPED, not copied (r06 profile: the fread copy was 9 % of the host stage): the parser touches the metadata and
   * each chunk once, the inflate reads straight from the page cache.  MAP_PRIVATE + read-only: a file that shrinks while it is
   * read is the one case a copy handled and a mapping does not (SIGBUS) - fast5 files are written once; NRV_HOST_MMAP=0 copies. */
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return fail(err, err_len, NRVH_E_IO, "cannot open the file");
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0 || sb.st_size >= ((off_t)1 << 36)) {
    close(fd);
    return fail(err, err_len, NRVH_E_IO, "cannot read the file");
  }
  const long fsz = (long)sb.st_size;
  static int use_mmap_s = -1;                           /* threads race to the same answer: relaxed atomics keep TSan quiet */
  int use_mmap = __atomic_load_n(&use_mmap_s, __ATOMIC_RELAXED);
  if (use_mmap < 0) {
    const char* e = getenv("NRV_HOST_MMAP");
    use_mmap = !(e && e[0] == '0');
    __atomic_store_n(&use_mmap_s, use_mmap, __ATOMIC_RELAXED);
  }
  uint8_t* file = 0;
  int mapped = 0;
  if (use_mmap) {
    void* m = mmap(0, (size_t)fsz, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) { file = (uint8_t*)m; mapped = 1; }
  }
  if (!file) {
    file = (uint8_t*)malloc((size_t)fsz);
    size_t got = 0;
    while (file && got < (size_t)fsz) {
      const ssize_t r = read(fd, file + got, (size_t)fsz - got);
      if (r <= 0) break;
      got += (size_t)r;
    }
    if (!file || got != (size_t)fsz) { close(fd); free(file); return fail(err, err_len, NRVH_E_IO, "cannot read the file"); }
  }
  close(fd);
  const int rc = load_fast5_image(file, fsz, group, subgroup, want_fastq, out, err, err_len);
  if (mapped) munmap(file, (size_t)fsz); else free(file);
  return rc;
}

/* ---- several files -> the arrays of ONE device call ------------------------------------------------------------------ */
void nrvh_free_bundle(nrvh_bundle* b) {
  if (!b) return;
  free(b->raw); free(b->starts); free(b->feat); free(b->bases); free(b->meta); free(b->status); free(b->fastq);
  free(b->fastq_off); free(b->errors);
  memset(b, 0, sizeof *b);
}

int nrvh_load_bundle(const char* const* paths, int n, const char* group, const char* subgroup, int want_fastq,
                     nrvh_bundle* out) {
  if (!paths || n < 0 || !group || !subgroup || !out) return NRVH_E_ARG;
  memset(out, 0, sizeof *out);
  nrvh_read* rd = (nrvh_read*)calloc((size_t)(n > 0 ? n : 1), sizeof(nrvh_read));
  out->n_files = n;
  out->meta = (double*)calloc((size_t)(n > 0 ? n : 1) * 4, sizeof(double));
  out->status = (int32_t*)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
  out->fastq_off = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
  out->errors = (char*)calloc((size_t)(n > 0 ? n : 1), NRVH_ERR_LEN);
  int rc = NRVH_E_IO;
  if (rd && out->meta && out->status && out->fastq_off && out->errors) {
    int64_t tr = 0, te = 0, tf = 0;
    for (int i = 0; i < n; ++i) {
      out->status[i] = nrvh_load_fast5(paths[i], group, subgroup, want_fastq, &rd[i], out->errors + (size_t)i * NRVH_ERR_LEN, NRVH_ERR_LEN);
      if (out->status[i] == NRVH_OK) { tr += rd[i].n_raw; te += rd[i].n_ev; tf += rd[i].fastq_len; ++out->n_ok; }
    }
    out->n_raw = tr; out->n_ev = te;
    out->raw = (int16_t*)malloc((size_t)(tr > 0 ? tr : 1) * 2);
    out->starts = (int32_t*)malloc((size_t)(te > 0 ? te : 1) * 4);
    out->feat = (float*)malloc((size_t)(te > 0 ? te : 1) * 24);
    out->bases = (char*)malloc((size_t)te + 1);
    out->fastq = (char*)malloc((size_t)tf + 1);
    if (out->raw && out->starts && out->feat && out->bases && out->fastq) {
      int64_t ro = 0, eo = 0, fo = 0;
      for (int i = 0; i < n; ++i) {
        out->fastq_off[i] = fo;
        if (out->status[i] != NRVH_OK) continue;
        const nrvh_read* r = &rd[i];
        memcpy(out->raw + ro, r->raw, (size_t)r->n_raw * 2);
        memcpy(out->starts + eo, r->starts, (size_t)r->n_ev * 4);
        memcpy(out->feat + eo * 6, r->feat, (size_t)r->n_ev * 24);
        memcpy(out->bases + eo, r->bases, (size_t)r->n_ev);
        if (r->fastq_len) memcpy(out->fastq + fo, r->fastq, (size_t)r->fastq_len);
        double* m = out->meta + (size_t)i * 4;
        m[0] = (double)r->n_raw; m[1] = (double)r->n_ev; m[2] = r->shift; m[3] = r->scale;
        ro += r->n_raw; eo += r->n_ev; fo += r->fastq_len;
        if (!r->fastq) out->fastq_off[i] = -1;          /* no Fastq record in that file (or not asked for) */
      }
      out->fastq_off[n] = fo;
      out->bases[te] = 0; out->fastq[tf] = 0;
      rc = NRVH_OK;
    }
  }
  if (rd) for (int i = 0; i < n; ++i) nrvh_free_read(&rd[i]);
  free(rd);
  if (rc != NRVH_OK) nrvh_free_bundle(out);
  return rc;
}

/* ---- calls of one read -> the revised read -> its output file --------------------------------------------------------
 * hoststage.merge_calls / expand_calls / revise_read (the decode of output_handeler.py:83, 104-122 with SURVEY 8a a16's
 * fix) + fasta_record / fastq_record (output_handeler.py:26-62, byte for byte, including the missing newline before '+')
 * + cli.write_read's temporary-then-rename.  qc: one Phred character per window (FASTQ) or NULL (FASTA). */
int nrvh_finish_read(const char* bases, int64_t n_ev, const int8_t* a1, const int8_t* a2, int64_t n_win, int T,
                     const uint8_t* qc, const char* name, const char* dst, int fastq, int64_t* n_written) {
  if (!bases || n_ev < 0 || n_win < 0 || (n_win > 0 && (!a1 || !a2)) || !name || !dst || T < 1) return NRVH_E_ARG;
  static const char lab[6] = {'D', '-', 'C', 'T', 'G', 'A'};
  const int64_t off = (T - 1) / 2;
  if (n_win > 0 && off + n_win > n_ev) return NRVH_E_ARG;
  const size_t nl = strlen(name);
  char* text = (char*)malloc(nl + 8 + 2 * (size_t)(n_ev + n_win) + 8);
  char* qual = fastq ? (char*)malloc((size_t)(n_ev + n_win) + 8) : 0;
  if (!text || (fastq && !qual)) { free(text); free(qual); return NRVH_E_IO; }
  size_t p = 0, q = 0;
  text[p++] = fastq ? '@' : '>';
  memcpy(text + p, name, nl); p += nl;
  text[p++] = '\n';
  const size_t seq0 = p;
  for (int64_t i = 0; i < off && i < n_ev; ++i) { text[p++] = bases[i]; if (qual) qual[q++] = '#'; }
  for (int64_t i = 0; i < n_win; ++i) {
    const int x = a1[i], y = a2[i] + 1;
    const char orig = bases[off + i];
    const int agree = x == y && x >= 2, dele = x == 0 && y >= 2, drop = x == 1 && y == 1;
    if (drop) continue;
    const int cx = x < 0 ? 0 : (x > 5 ? 5 : x), cy = y < 0 ? 0 : (y > 5 ? 5 : y);
    text[p++] = agree ? lab[cx] : orig;
    if (qual) qual[q++] = (char)(qc ? qc[i] : '#');
    if (dele) { text[p++] = lab[cy]; if (qual) qual[q++] = (char)(qc ? qc[i] : '#'); }
  }
  for (int64_t i = n_win > 0 ? off + n_win : (off < n_ev ? off : n_ev); i < n_ev; ++i) { text[p++] = bases[i]; if (qual) qual[q++] = '#'; }
  const int64_t nseq = (int64_t)(p - seq0);
  if (fastq) {
    text = (char*)realloc(text, p + 2 + q + 1);
    if (!text) { free(qual); return NRVH_E_IO; }
    text[p++] = '+'; text[p++] = '\n';
    memcpy(text + p, qual, q); p += q;
  }
  free(qual);
  /* the finishers are threads of one process and two reads can map to one dst (the stem ends at the first '.'): the
   * temporary is unique per CALL, so the writes never interleave and the last rename wins whole */
  static unsigned long tmp_serial;
  char tmp[4200];
  if (snprintf(tmp, sizeof tmp, "%s.tmp%ld_%lu", dst, (long)getpid(), __atomic_add_fetch(&tmp_serial, 1, __ATOMIC_RELAXED)) >= (int)sizeof tmp) { free(text); return NRVH_E_ARG; }
  FILE* fp = fopen(tmp, "wb");
  if (!fp) { free(text); return NRVH_E_IO; }
  const int ok = fwrite(text, 1, p, fp) == p;
  const int ok2 = fclose(fp) == 0;
  free(text);
  if (!ok || !ok2 || rename(tmp, dst) != 0) { remove(tmp); return NRVH_E_IO; }
  if (n_written) *n_written = nseq;
  return NRVH_OK;
}

/* The reads of ONE device call at once: bases / a1 / a2 / qc are the call's concatenated arrays (read r has ev_len[r]
 * bases; its window i is window e0 + i of the call, e0 the bases in front of it; it has max(ev_len[r] - T, 0) windows).
 * status[r] = NRVH_OK or the error of that read alone (the others are still written). */
int nrvh_finish_bundle(const char* bases, const int64_t* ev_len, int n_reads, const int8_t* a1, const int8_t* a2,
                       int64_t n_win_total, int T, const uint8_t* qc, const char* const* names, const char* const* dsts,
                       int fastq, int64_t* n_written, int32_t* status) {
  if (!bases || !ev_len || n_reads < 0 || !names || !dsts || !status || T < 1) return NRVH_E_ARG;
  int64_t e0 = 0;
  for (int r = 0; r < n_reads; ++r) {
    const int64_t el = ev_len[r], n = el - T > 0 ? el - T : 0;
    if (el < 0 || (n > 0 && e0 + n > n_win_total)) { status[r] = NRVH_E_ARG; e0 += el > 0 ? el : 0; continue; }
    int64_t nw = 0;
    status[r] = nrvh_finish_read(bases + e0, el, a1 ? a1 + e0 : 0, a2 ? a2 + e0 : 0, n, T, qc ? qc + e0 : 0, names[r], dsts[r],
                                 fastq, &nw);
    if (n_written) n_written[r] = nw;
    e0 += el;
  }
  return NRVH_OK;
}
