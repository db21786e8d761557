/*
 * ref_capture.c -- TEST INFRASTRUCTURE (oracle side). Not part of the product.
 *
 * Full-precision stdout for the *real* reference `gortt`.  The reference prints
 * every number with "%f" (6 decimals; /root/reference/gortt.c:310-327), which
 * cannot resolve a 1e-5 *relative* error on reflectances ~0.005.  oracle/Makefile
 * therefore builds a second copy of the reference, straight from the sources
 * under /root/reference, with `-Dprintf=gort_ref_capture_printf`: every printf()
 * call of the reference's own main() lands here, and every floating-point
 * conversion is re-emitted as "%.17g" (round-trip exact).  Nothing else about the
 * reference program changes - same main(), same angle normalisation, same flags.
 *
 * This file contains no reference source; it only re-implements the tiny subset of
 * printf format handling that the reference uses (%s %d %f %lf %0.40f).
 */
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

int gort_ref_capture_printf(const char *fmt, ...)
{
    va_list ap;
    int n = 0;
    va_start(ap, fmt);
    for (const char *c = fmt; *c; ++c) {
        if (*c != '%') { fputc(*c, stdout); ++n; continue; }
        ++c;
        if (*c == '%') { fputc('%', stdout); ++n; continue; }
        /* skip flags / width / precision / length */
        while (*c && strchr("0123456789.-+ #lh", *c)) ++c;
        switch (*c) {
        case 'f': case 'g': case 'e':
            n += fprintf(stdout, "%.17g", va_arg(ap, double));
            break;
        case 'd': case 'i':
            n += fprintf(stdout, "%d", va_arg(ap, int));
            break;
        case 's':
            n += fprintf(stdout, "%s", va_arg(ap, char *));
            break;
        default:
            /* the reference uses nothing else; make it loud if that changes */
            fprintf(stderr, "gort_ref_capture_printf: unhandled conversion %%%c\n", *c);
            break;
        }
    }
    va_end(ap);
    return n;
}
