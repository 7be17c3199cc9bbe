#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "pgv_hip.h"
#include "pgv_host.h"
static uint64_t lcg = 12345;
static uint32_t urand(void){ lcg = lcg*6364136223846793005ull+1442695040888963407ull; return (uint32_t)(lcg>>33);}
int main(int argc,char**argv){
  int N = argc>1?atoi(argv[1]):200000, DIM=8, M=16, EFC=64, B=argc>2?atoi(argv[2]):1024;
  float*data=malloc(sizeof(float)*(size_t)N*DIM);
  for(size_t i=0;i<(size_t)N*DIM;i++) data[i]=(float)(urand()%100000)/1000.0f;
  pgv_ctx*ctx; pgv_hnsw*mirror; pgv_hnsw_built built;
  pgv_ctx_create(0,NULL,&ctx);
  pgv_hnsw_upload(ctx,PGV_L2SQ,PGV_F32,DIM,data,N,&mirror);
  int rc=pgv_host_hnsw_build(mirror,PGV_F32,DIM,data,N,M,EFC,NULL,B,&built);
  printf("rc %d batches %ld pairs %ld\n",rc,(long)built.batches,(long)built.device_pairs);
  const char*names[8]={"search","pairs","select","records","update","patch","pairlist","free"};
  for(int i=0;i<8;i++) printf("%-9s %.3f\n",names[i],built.phase_secs[i]);
  return 0;}
