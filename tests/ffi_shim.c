/* ffi_shim.c -- TEST INFRASTRUCTURE: a C caller with the call shape of the reference's Rust host.
 *
 * The reference binds this library from Rust (/root/reference/src/ecoz2_lib/mod.rs); there is no Rust toolchain in this image,
 * so this file restates, in C, exactly what that caller does at the ABI -- tests/test_ffi_shim.py builds it with gcc (no HIP,
 * no C++ runtime, none of this repo's headers) and drives config 1's golden files through it:
 *   - the entry points are declared WITHOUT a return value, as the Rust `extern "C"` block declares them
 *     (mod.rs:96-122: `fn ecoz2_vq_learn(...)` etc. return `()`): whatever the library returns in eax is ignored;
 *   - `#[repr(C)] struct Ecoz2ObserverRef { ref_id: c_int }` (mod.rs:51-54) is handed to the library as `target` and comes
 *     back as argument 0 of the callback `extern "C" fn(*mut Ecoz2ObserverRef, c_int, c_double, c_double, c_double)`
 *     (mod.rs:103-104, 241-250), which reads ref_id through it like Ecoz2ObserverRef::step does (mod.rs:61-69);
 *   - the learn calls run on a helper thread with a 128 MiB stack (mod.rs:271-275: "thread needed to increment stack size");
 *   - file names travel as an array of borrowed `*const c_char` built from heap strings (to_vec_of_ptr_const_c_char,
 *     mod.rs:203 / 278 / 330) that the caller frees only after the call returned (here: never -- leaked, as CString::into_raw
 *     arrays are);
 *   - the base codebook of `-B` is passed as `String::as_ptr()` (mod.rs:295): a buffer that is NOT NUL-terminated by the
 *     caller -- mode learn_base puts the bytes "\x01garbage" + NUL behind the name, as a heap neighbour might.
 * Prints one line per callback, doubles as C99 hex floats:  STEP <ref_id> <M> <avg> <sigma> <inertia>
 *
 *   ffi_shim version
 *   ffi_shim learn <P> <eps> <class> <prd>...
 *   ffi_shim learn_base <base.cbook> <eps> <prd>...
 *   ffi_shim quantize <codebook> <show_filenames> <prd>...
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct Ecoz2ObserverRef {
    int ref_id;
};
typedef void (*learn_cb_t)(struct Ecoz2ObserverRef *, int, double, double, double);

extern const char *ecoz2_version(void);
extern void ecoz2_vq_learn(int prediction_order, double epsilon, const char *codebook_class_name,
                           const char *const *predictor_filenames, int num_predictors, struct Ecoz2ObserverRef *target,
                           learn_cb_t callback);
extern void ecoz2_vq_learn_using_base_codebook(const char *base_codebook, double epsilon, const char *const *predictor_filenames,
                                               int num_predictors, struct Ecoz2ObserverRef *target, learn_cb_t callback);
extern void ecoz2_vq_quantize(const char *nom_raas, const char *const *predictor_filenames, int num_predictors, int show_filenames);

static void c_vq_learn_callback(struct Ecoz2ObserverRef *target, int m, double avg_distortion, double sigma, double inertia)
{
    printf("STEP %d %d %a %a %a\n", target->ref_id, m, avg_distortion, sigma, inertia);
    fflush(stdout);
}

struct job {
    int argc;
    char **argv;
};

/* heap copies of the names and a heap array of pointers to them, never freed */
static const char *const *leak_names(char **names, int n)
{
    const char **v = (const char **)malloc((size_t)(n > 0 ? n : 1) * sizeof *v);
    for (int i = 0; i < n; ++i) v[i] = strdup(names[i]);
    return v;
}

static void *learn_thread(void *arg)
{
    struct job *j = (struct job *)arg;
    /* deep frames below the call, as the C implementation the stack was enlarged for would have them */
    volatile char pad[1 << 20];
    pad[0] = pad[sizeof pad - 1] = 1;
    struct Ecoz2ObserverRef *obs = (struct Ecoz2ObserverRef *)malloc(sizeof *obs); /* Box::new(observer) */
    obs->ref_id = 4242;
    if (strcmp(j->argv[1], "learn") == 0) {
        const int n = j->argc - 5;
        ecoz2_vq_learn(atoi(j->argv[2]), atof(j->argv[3]), strdup(j->argv[4]), leak_names(j->argv + 5, n), n, obs, c_vq_learn_callback);
    } else {
        const int n = j->argc - 4;
        /* String::as_ptr(): the name's bytes without a terminator of their own; what follows is whatever the heap holds */
        const size_t len = strlen(j->argv[2]);
        char *raw = (char *)malloc(len + 16);
        memcpy(raw, j->argv[2], len);
        memcpy(raw + len, "\x01garbage", 9); /* (9: with the NUL a C string walk eventually needs to stop at) */
        ecoz2_vq_learn_using_base_codebook(raw, atof(j->argv[3]), leak_names(j->argv + 4, n), n, obs, c_vq_learn_callback);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "version") == 0) {
        printf("VERSION %s\n", ecoz2_version());
        return 0;
    }
    if (argc >= 6 && strcmp(argv[1], "learn") == 0 || argc >= 5 && strcmp(argv[1], "learn_base") == 0) {
        struct job j = {argc, argv};
        pthread_attr_t at;
        pthread_t th;
        pthread_attr_init(&at);
        if (pthread_attr_setstacksize(&at, (size_t)128 * 1024 * 1024) != 0) return 3; /* thread::Builder::stack_size */
        if (pthread_create(&th, &at, learn_thread, &j) != 0) return 4;
        pthread_join(th, NULL); /* child.join().unwrap() */
        printf("DONE\n");
        return 0;
    }
    if (argc >= 5 && strcmp(argv[1], "quantize") == 0) {
        const int n = argc - 4;
        ecoz2_vq_quantize(strdup(argv[2]), leak_names(argv + 4, n), n, atoi(argv[3])); /* (on the caller's own thread: mod.rs:325-342) */
        printf("DONE\n");
        return 0;
    }
    fprintf(stderr, "usage: ffi_shim version | learn P eps class prd... | learn_base base.cbook eps prd... | quantize codebook show prd...\n");
    return 2;
}
