/* omp.h for the ThreadSanitizer build of tests/test_hnsw_build_tsan_cpu.py: without -fopenmp the pragmas are ignored and
 * the team is one thread, so that what the sanitizer sees are the build's own helper threads (libgomp's barriers are
 * invisible to it and would read as races).  TEST INFRASTRUCTURE. */
#pragma once
static inline int omp_get_max_threads(void){return 1;}
static inline int omp_get_thread_num(void){return 0;}
static inline int omp_get_num_threads(void){return 1;}
static inline int omp_get_num_procs(void){return 1;}
static inline void omp_set_num_threads(int n){(void)n;}
static inline double omp_get_wtime(void){return 0;}
static inline int omp_in_parallel(void){return 0;}
