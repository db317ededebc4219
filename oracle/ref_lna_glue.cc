// ref_lna_glue.cc -- extern "C" shim over the reference decoder's own LNA
// reader (decoder/src/LnaReaderCircular.cc, compiled in place into
// oracle/_ref/liblna_ref.so by oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
// It pins the consumer side of the LNA format: files written by the oracle
// and by the engine are opened with the reader the reference recogniser uses
// and its log_prob() view is compared with ours.
#include <cstdio>
#include <cstdlib>

#include "LnaReaderCircular.hh"

extern "C" {

// Reads every frame of `path` through go_to()/log_prob().  `out` receives
// frames x num_models floats (up to max_frames).  `order` selects the access
// pattern: 0 = forward, 1 = forward with a look-back of `buf_size - 1` frames
// after each step (exercises the circular buffer).  Returns the number of
// frames, *num_models is set from the header.
int ref_lna_read(const char *path, int buf_size, int order, float *out,
                 int max_frames, int *num_models)
{
  LnaReaderCircular r;
  r.open_file(path, buf_size);
  *num_models = r.num_models();
  const int S = r.num_models();
  int t = 0;
  while (t < max_frames && r.go_to(t)) {
    if (order == 1 && t >= buf_size - 1) {
      // revisit the oldest frame the buffer must still hold, then return
      const int back = t - (buf_size - 1);
      if (!r.go_to(back)) break;
      for (int s = 0; s < S; s++)
        if (out[(size_t)back * S + s] != r.log_prob(s)) { r.close(); return -2; }
      if (!r.go_to(t)) break;
    }
    for (int s = 0; s < S; s++) out[(size_t)t * S + s] = r.log_prob(s);
    t++;
  }
  r.close();
  return t;
}

}  // extern "C"
