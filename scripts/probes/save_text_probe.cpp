#include "../../flashpca_amd/csrc/plink_io.hpp"
#include <chrono>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>
using namespace fpca;
int main(int argc, char**argv){
  const uint64_t N=500000,k=20,P=100000;
  std::vector<double> U(N*k),V(P*k);
  std::mt19937_64 g(1); std::normal_distribution<double> d(0,1e-3);
  for(auto&x:U)x=d(g); for(auto&x:V)x=d(g);
  std::vector<std::string> rn(N),cn(k+1,"c"),rs(P);
  for(uint64_t i=0;i<N;i++)rn[i]="F"+std::to_string(i)+"\tI"+std::to_string(i);
  for(uint64_t i=0;i<P;i++)rs[i]="rs"+std::to_string(i)+"\tA";
  auto t0=std::chrono::steady_clock::now();
  auto lap=[&](const char*w){auto t=std::chrono::steady_clock::now(); printf("%-40s %.1f ms\n",w,std::chrono::duration<double>(t-t0).count()*1e3); t0=t;};
  printf("usable cpus %u\n", usable_cpus());
  unsigned share = argc>1? atoi(argv[1]) : 0;
  save_text(U.data(),N,k,cn,rn,"a.txt",7,share); lap("sequential: U");
  save_text(U.data(),N,k,cn,rn,"b.txt",7,share); lap("sequential: Px");
  save_text(V.data(),P,k,cn,rs,"c.txt",7,share); lap("sequential: V");
  unsigned sh = argc>2? atoi(argv[2]) : std::max(2u, usable_cpus()/3);
  std::thread a([&]{save_text(U.data(),N,k,cn,rn,"a.txt",7,sh);}), b([&]{save_text(U.data(),N,k,cn,rn,"b.txt",7,sh);}), c([&]{save_text(V.data(),P,k,cn,rs,"c.txt",7,sh);});
  a.join();b.join();c.join(); lap("concurrent x3");
}
