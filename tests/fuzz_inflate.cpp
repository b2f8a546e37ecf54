// tests/fuzz_inflate.cpp -- host/mcom_inflate.cpp under AddressSanitizer + UBSan (tests/test_fastq.py builds and runs it on the CPU): every level and
// strategy of zlib over five kinds of data decoded into heap buffers of EXACTLY the needed size and of sizes around the fast loop's margins, the same
// streams decoded in pieces of odd sizes (mcom_inflate_run with the 32 KB history in front of every piece), thousands of mutated members (bit
// flips, overwritten bytes, truncations) and random bytes behind a valid header.  Any read or write outside a buffer ends the run.
#include "mcom_inflate.hpp"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
int main(){
  std::mt19937_64 rng(12345);
  // corpus: several kinds of data, several levels/strategies
  std::vector<std::vector<unsigned char>> datas;
  { std::vector<unsigned char> d(200000); for(auto&x:d) x="ACGT"[rng()&3]; datas.push_back(d); }
  { std::vector<unsigned char> d(150000); for(size_t i=0;i<d.size();++i) d[i]=(unsigned char)((i*7)%251); datas.push_back(d); }
  { std::vector<unsigned char> d(100000); for(auto&x:d) x=(unsigned char)rng(); datas.push_back(d); }
  { std::vector<unsigned char> d(300000,'I'); for(size_t i=0;i<d.size();i+=151) d[i]='\n'; datas.push_back(d); }
  { std::string s; for(int i=0;i<2000;++i){ char b[400]; int n=snprintf(b,sizeof b,"@r%d\n",i); s.append(b,n); for(int j=0;j<150;++j) s.push_back("ACGT"[rng()&3]); s+="\n+\n"; for(int j=0;j<150;++j) s.push_back("FFFF:,#"[rng()%7]); s+="\n"; } datas.emplace_back(s.begin(),s.end()); }
  long runs=0, ok=0, room=0, err=0;
  for(auto&d:datas) for(int lvl: {0,1,6,9}) for(int strat: {Z_DEFAULT_STRATEGY,Z_FIXED,Z_HUFFMAN_ONLY,Z_RLE,Z_FILTERED}){
    z_stream z; memset(&z,0,sizeof z); deflateInit2(&z,lvl,Z_DEFLATED,31,8,strat);
    std::vector<unsigned char> c(deflateBound(&z,d.size())+64); z.next_in=d.data(); z.avail_in=d.size(); z.next_out=c.data(); z.avail_out=c.size(); deflate(&z,Z_FINISH); c.resize(z.total_out); deflateEnd(&z);
    // exact decode with exact-size heap buffers (ASan sees any overrun)
    for(size_t cap: {d.size(), d.size()+1, d.size()+299, d.size()+301, d.size()/2, (size_t)0, (size_t)1}){
      unsigned char*in=(unsigned char*)malloc(c.size()); memcpy(in,c.data(),c.size());
      unsigned char*out=(unsigned char*)malloc(cap?cap:1); size_t u=0,n=0;
      int rc=mcom_gunzip_member(in,c.size(),out,cap,&u,&n); ++runs;
      if(cap>=d.size()){ if(rc!=0||n!=d.size()||memcmp(out,d.data(),n)){ printf("MISMATCH lvl %d strat %d cap %zu rc %d\n",lvl,strat,cap,rc); return 1;} ++ok; } else { if(rc!=1){ printf("expected ROOM got %d (cap %zu of %zu)\n",rc,cap,d.size()); return 1;} ++room; }
      free(in); free(out);
    }
    // streamed decode in pieces of odd sizes with history copy
    { mcom_inflate_stream st; // skip 10-byte header
      unsigned char*in=(unsigned char*)malloc(c.size()); memcpy(in,c.data(),c.size());
      mcom_inflate_begin(&st,in+10,c.size()-10);
      std::vector<unsigned char> all; size_t piece=1+rng()%70000; std::vector<unsigned char> hist(32768); size_t hn=0; int rc;
      do { unsigned char*buf=(unsigned char*)malloc(32768+piece); memcpy(buf+32768-hn,hist.data()+32768-hn,hn); size_t n=0; rc=mcom_inflate_run(&st,buf+32768,piece,hn,&n); if(rc<0){printf("stream err %d\n",rc);return 1;} all.insert(all.end(),buf+32768,buf+32768+n);
           if(n>=32768){memcpy(hist.data(),buf+32768+n-32768,32768);hn=32768;} else {memmove(hist.data(),hist.data()+n,32768-n); memcpy(hist.data()+32768-n,buf+32768,n); hn=std::min<size_t>(32768,hn+n);} free(buf); } while(rc==1);
      mcom_inflate_end(&st); free(in);
      if(all.size()!=d.size()||memcmp(all.data(),d.data(),d.size())){ printf("STREAM MISMATCH lvl %d strat %d piece %zu: %zu vs %zu\n",lvl,strat,piece,all.size(),d.size()); return 1;} }
    // mutations
    for(int t=0;t<150;++t){ std::vector<unsigned char> m=c; int k=1+rng()%3; for(int q=0;q<k;++q){ size_t at=10+rng()%(m.size()-10); int how=rng()%3; if(how==0) m[at]^=1u<<(rng()%8); else if(how==1) m[at]=(unsigned char)rng(); else m.resize(at+1);} 
      unsigned char*in=(unsigned char*)malloc(m.size()); memcpy(in,m.data(),m.size()); size_t cap=rng()%(2*d.size()+2);
      unsigned char*out=(unsigned char*)malloc(cap?cap:1); size_t u=0,n=0; int rc=mcom_gunzip_member(in,m.size(),out,cap,&u,&n); ++runs; if(rc) ++err; free(in); free(out); }
  }
  // pure garbage
  for(int t=0;t<20000;++t){ size_t len=1+rng()%4000; unsigned char*in=(unsigned char*)malloc(len+10); memcpy(in,"\x1f\x8b\x08\x00\0\0\0\0\0\x03",10); for(size_t i=0;i<len;++i) in[10+i]=(unsigned char)rng(); size_t cap=rng()%200000; unsigned char*out=(unsigned char*)malloc(cap?cap:1); size_t u,n; mcom_gunzip_member(in,len+10,out,cap,&u,&n); ++runs; free(in); free(out);} 
  printf("runs %ld exact ok %ld room %ld mutated rejected %ld\n",runs,ok,room,err); return 0; }
