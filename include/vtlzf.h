/* vtlzf.h — C ABI of libvtlzf.so: the LZF block codec behind vlatouch/h5lite.py (host side, plain C: csrc/vt_lzf.c).
 * Replaces what h5py's bundled HDF5 filter 32000 ("lzf") does for the reference's episode files
 * (VLA/data/franka_data/4_convert_to_hdf5.py:41-44 `compression='lzf'`; read back in residual_controller/controller_dataset.py:71-170).
 * Format: Marc Lehmann's liblzf block stream (control byte < 32: literal run; else back reference, see vt_lzf.c). */
#ifndef VTLZF_H
#define VTLZF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* -> number of bytes written to `out` (must equal the caller's expected size), or -1 on a malformed / overflowing stream. */
long vt_lzf_decompress(const uint8_t* in, long in_len, uint8_t* out, long out_len);
/* -> compressed size, or 0 when the result would not fit in out_cap (HDF5 then stores the chunk raw, filter mask bit set). */
long vt_lzf_compress(const uint8_t* in, long in_len, uint8_t* out, long out_cap);
#ifdef __cplusplus
}
#endif
#endif
