#include <cstdio>
#include <vector>
#include <cstdint>
#include "../../include/pzg.h"
int main(){
  pzg_ctx* ctx=nullptr; int rc=pzg_init(0,&ctx); printf("init rc=%d\n",rc); if(rc) return 1;
  std::vector<uint8_t> b(100000, 7); uint32_t out=0;
  rc=pzg_adler32(ctx,b.data(),b.size(),1,&out,0); printf("adler rc=%d out=%08x err=%s\n",rc,out,pzg_last_error(ctx));
  fflush(stdout);
  uint8_t z[]={0x78,0x9c,0x03,0x00,0x00,0x00,0x00,0x01}; uint8_t o[16]; uint64_t ol=0,iu=0; int32_t st=-1; uint32_t d[2];
  rc=pzg_decompress(ctx,z,sizeof z,o,16,&ol,&st,d,&iu); printf("dec rc=%d st=%d ol=%llu err=%s\n",rc,st,(unsigned long long)ol,pzg_last_error(ctx));
  pzg_shutdown(ctx); return 0; }
