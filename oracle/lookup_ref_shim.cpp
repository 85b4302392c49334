// lookup_ref_shim.cpp — extern "C" face for the reference's OWN LookupTable, compiled from the sources where they lie
// under /root/reference (scene_flow_clusterer/include/lookup_table.h, src/lookup_table.cpp).  Nothing of the reference
// is copied into this repository: this file only includes the header by name at build time.  Output goes to
// oracle/_ref/ (git-ignored, but shipped to the GPU box).  TEST INFRASTRUCTURE ONLY.
#include <cstddef>
#include "lookup_table.h"

extern "C" {
void *ref_lut_create(long size) { LookupTable *l = new LookupTable((size_t)size); l->reset(); return l; }
void ref_lut_destroy(void *l) { delete (LookupTable *)l; }
void ref_lut_reset(void *l) { ((LookupTable *)l)->reset(); }
int ref_lut_add_label(void *l) { return ((LookupTable *)l)->addLabel(); }
void ref_lut_link(void *l, int a, int b) { ((LookupTable *)l)->link(a, b); }
int ref_lut_lookup(void *l, int s) { return ((LookupTable *)l)->lookup(s); }
}
