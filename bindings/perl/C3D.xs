/* C3D.xs — Perl XS binding of the libc3d C ABI (include/c3d.h): the in-process FFI through which the
 * Perl driver (bin/chromosome3D_amd.pl) reaches the HIP kernels, replacing the reference's
 * system("./job.sh") -> `cns_solve < dgsa.inp` process boundary (chromosome3D.pl:254-289).
 *
 *   my $r = C3D::solve($matrix, $outdir, $ID, $models, $K, $alpha, $seed, $device, $embed);
 *       IF2dist_new + dist2rr + carr2tbl (:87-89) on the GPU/host -> <ID>.dist/.rr, contact.tbl,
 *       build_models (:104) -> <ID>_<k>.pdb with REMARK noe; returns a hash ref
 *       { n, restraints, steps, ms, e_noe => [..], spearman => [..] }
 *   my $rho = C3D::score($matrix, $pdb, $range);      # spearman_IF_pdb.pl:26-70
 *   C3D::last_error()
 */
#define PERL_NO_GET_CONTEXT
#include "EXTERN.h"
#include "perl.h"
#include "XSUB.h"

#include <stdlib.h>
#include <string.h>

#include "c3d.h"

#define C3D_TRY(call) do { if ((call) != C3D_OK) goto fail; } while (0)

MODULE = C3D		PACKAGE = C3D

PROTOTYPES: DISABLE

BOOT:
    /* a device exception should reach stderr in the runtime's own words, not end in ROCr's GPU-core-dump helper (INTEGRATION.md "Device
     * exceptions"); set at module load, before the first HIP call of this perl; a user's own setting wins */
    setenv("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1", 0);

const char*
last_error()
    CODE:
        RETVAL = c3d_last_error();
    OUTPUT:
        RETVAL

const char*
version()
    CODE:
        RETVAL = c3d_version();
    OUTPUT:
        RETVAL

int
device_count()
    CODE:
        RETVAL = c3d_device_count();
    OUTPUT:
        RETVAL

int
set_sequence(seq1)
        const char* seq1
    CODE:
        RETVAL = c3d_set_residue_sequence(seq1);
    OUTPUT:
        RETVAL

SV*
solve(matrix_path, out_dir, id, models, K, alpha, seed, device, embed)
        const char* matrix_path
        const char* out_dir
        const char* id
        int models
        double K
        double alpha
        UV seed
        int device
        int embed
    PREINIT:
        c3d_ctx* ctx = NULL;
        double* IF = NULL;
        int32_t* d10 = NULL;
        float* xyz = NULL;
        double* en = NULL;
        double* rho = NULL;
        int n = 0, R = 0, k, nst;
        c3d_model model;
        c3d_fire_params fire;
        c3d_stage* stages = NULL;
        char path[4096], p2[4096], p3[4096], name[512];
        double ms = 0;
        long steps = 0, launches = 0;
        HV* out;
        AV *ae, *ar;
    CODE:
        C3D_TRY(c3d_create(device, &ctx));
        c3d_default_model(&model);
        C3D_TRY(c3d_set_model(ctx, &model));
        C3D_TRY(c3d_parse_if_file(matrix_path, &IF, &n));
        C3D_TRY(c3d_set_if_matrix(ctx, IF, n, alpha, K));
        d10 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * n);
        C3D_TRY(c3d_get_dist10(ctx, d10));
        snprintf(path, sizeof path, "%s/%s.dist", out_dir, id);
        snprintf(p2, sizeof p2, "%s/%s.rr", out_dir, id);
        snprintf(p3, sizeof p3, "%s/contact.tbl", out_dir);
        C3D_TRY(c3d_write_front_half(d10, n, model.min_sep, path, p2, p3, &R));
        nst = c3d_default_schedule(NULL, 0, 3000);
        stages = (c3d_stage*)malloc(sizeof(c3d_stage) * nst);
        c3d_default_schedule(stages, nst, 3000);
        c3d_default_fire(&fire);
        C3D_TRY(c3d_set_schedule(ctx, stages, nst, &fire, 1e-2f, 250));
        C3D_TRY(c3d_init_replicas(ctx, models, (uint64_t)seed, 0));
        if (embed) C3D_TRY(c3d_embed_replicas(ctx, 50));
        C3D_TRY(c3d_run(ctx));
        xyz = (float*)malloc(sizeof(float) * 3 * (size_t)n * models);
        en = (double*)malloc(sizeof(double) * 3 * models);
        rho = (double*)malloc(sizeof(double) * models);
        C3D_TRY(c3d_get_coords(ctx, xyz));
        C3D_TRY(c3d_get_energies(ctx, en));
        C3D_TRY(c3d_score_replicas(ctx, IF, 3, NULL, NULL, rho));      /* on the device, from the resident coordinates (K6): 3 ms instead of ~35 on the host at N = 455 */
        for (k = 0; k < models; ++k) {
            snprintf(name, sizeof name, "%s_%d.pdb", id, k + 1);
            snprintf(path, sizeof path, "%s/%s", out_dir, name);
            C3D_TRY(c3d_write_pdb(path, xyz + (size_t)k * n * 3, n, en[3 * k], en[3 * k + 1], en[3 * k + 2], name));
        }
        c3d_last_timing(ctx, &ms, &steps, &launches);
        out = newHV();
        ae = newAV();
        ar = newAV();
        for (k = 0; k < models; ++k) { av_push(ae, newSVnv(en[3 * k])); av_push(ar, newSVnv(rho[k])); }
        (void)hv_stores(out, "n", newSViv(n));
        (void)hv_stores(out, "restraints", newSViv(R));
        (void)hv_stores(out, "steps", newSViv(steps));
        (void)hv_stores(out, "ms", newSVnv(ms));
        (void)hv_stores(out, "e_noe", newRV_noinc((SV*)ae));
        (void)hv_stores(out, "spearman", newRV_noinc((SV*)ar));
        free(stages); free(d10); free(xyz); free(en); free(rho);
        c3d_free(IF);
        c3d_destroy(ctx);
        RETVAL = newRV_noinc((SV*)out);
        goto done;
      fail:
        free(stages); free(d10); free(xyz); free(en); free(rho);
        if (IF) c3d_free(IF);
        if (ctx) c3d_destroy(ctx);
        croak("C3D::solve: %s", c3d_last_error());
      done:
        ;
    OUTPUT:
        RETVAL

double
score(matrix_path, pdb_path, range)
        const char* matrix_path
        const char* pdb_path
        int range
    PREINIT:
        double* IF = NULL;
        float* xyz = NULL;
        int n = 0, m = 0;
        double rho = 0;
    CODE:
        if (c3d_parse_if_file(matrix_path, &IF, &n) != C3D_OK) croak("C3D::score: %s", c3d_last_error());
        if (c3d_read_pdb_ca(pdb_path, &xyz, &m) != C3D_OK) { c3d_free(IF); croak("C3D::score: %s", c3d_last_error()); }
        if (m != n) { c3d_free(IF); c3d_free(xyz); croak("C3D::score: mismatch in size! %d CA atoms vs %d x %d matrix", m, n, n); }
        if (c3d_spearman_if_dist(IF, xyz, n, range, &rho) != C3D_OK) { c3d_free(IF); c3d_free(xyz); croak("C3D::score: %s", c3d_last_error()); }
        c3d_free(IF);
        c3d_free(xyz);
        RETVAL = rho;
    OUTPUT:
        RETVAL

void
assess(tbl_path, relax, viol_path, ...)
        const char* tbl_path
        double relax
        const char* viol_path
    PREINIT:
        int32_t *ri = NULL, *rj = NULL, *rt = NULL;
        int R = 0, k, nm;
        int* sats = NULL;
        double* devs = NULL;
    PPCODE:
        /* count_satisfied_tbl_rows + sum_noe_dev (chromosome3D.pl:447-485, :581-600) of the model files given after viol_path, in that order:
           returns (count, total, sum_dev) per model, flattened; viol_path "" = no violation table.  (The results are pushed after the loop:
           a push inside it would overwrite the argument slots still to be read.) */
        nm = items - 3;
        if (c3d_read_tbl(tbl_path, &ri, &rj, &rt, &R) != C3D_OK) croak("C3D::assess: %s", c3d_last_error());
        Newx(sats, nm > 0 ? nm : 1, int);
        Newx(devs, nm > 0 ? nm : 1, double);
        for (k = 0; k < nm; ++k) {
            const char* pdb_path = SvPV_nolen(ST(k + 3));
            float* xyz = NULL;
            int n = 0, rc;
            sats[k] = 0; devs[k] = 0;
            if (c3d_read_pdb_ca(pdb_path, &xyz, &n) != C3D_OK) { c3d_free(ri); c3d_free(rj); c3d_free(rt); Safefree(sats); Safefree(devs); croak("C3D::assess: %s", c3d_last_error()); }
            rc = viol_path[0] ? c3d_write_violations(xyz, n, R, ri, rj, rt, relax, pdb_path, tbl_path, viol_path, &sats[k], &devs[k])
                              : c3d_assess(xyz, n, R, ri, rj, rt, relax, &sats[k], &devs[k]);
            c3d_free(xyz);
            if (rc != C3D_OK) { c3d_free(ri); c3d_free(rj); c3d_free(rt); Safefree(sats); Safefree(devs); croak("C3D::assess: %s", c3d_last_error()); }
        }
        c3d_free(ri); c3d_free(rj); c3d_free(rt);
        EXTEND(SP, 3 * nm);
        for (k = 0; k < nm; ++k) {
            mPUSHi(sats[k]);
            mPUSHi(R);
            mPUSHn(devs[k]);
        }
        Safefree(sats); Safefree(devs);
