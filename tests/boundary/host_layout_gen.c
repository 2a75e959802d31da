/* Prints the layout of the HOST layer's types (host/tsp_model.h) as #defines: the header the caller compiled against
 * the reference's own headers (ref_caller.c) statically asserts against.  Test infrastructure. */
#include <stddef.h>
#include <stdio.h>
#include "tsp_model.h"

#define P(ret, name, args) static ret (*const p_##name) args = name;
#include "protos.inc"
#undef P

int main(void)
{
    /* (the typed pointers above are the prototype check of the host header; keep them referenced) */
    const void *keep[] = {
#define P(ret, name, args) (const void *)p_##name,
#include "protos.inc"
#undef P
        0};
    (void)keep;
#define S(t) printf("#define H_SIZEOF_%s %zu\n", #t, sizeof(t));
#define F(t, f) printf("#define H_OFF_%s_%s %zu\n", #t, #f, offsetof(t, f));
#define V(x) printf("#define H_VAL_%s %ld\n", #x, (long)(x));
#include "layout_items.inc"
    return 0;
}
