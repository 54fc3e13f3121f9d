#include "../../flashpca_amd/csrc/plink_io.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <random>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
using namespace fpca;
int main(){
  std::string blob(140<<20,'x');
  auto t0=std::chrono::steady_clock::now();
  { std::ofstream o("/tmp/w1.bin", std::ios::binary); for(size_t i=0;i<blob.size();i+=1<<20) o.write(blob.data()+i,1<<20); o.close(); }
  auto t1=std::chrono::steady_clock::now(); printf("ofstream 140MB: %.1f ms\n", std::chrono::duration<double>(t1-t0).count()*1e3);
  t0=std::chrono::steady_clock::now();
  { int fd=open("/tmp/w2.bin",O_WRONLY|O_CREAT|O_TRUNC,0644); for(size_t i=0;i<blob.size();i+=1<<20) (void)!write(fd,blob.data()+i,1<<20); close(fd);}
  t1=std::chrono::steady_clock::now(); printf("write() 140MB: %.1f ms\n", std::chrono::duration<double>(t1-t0).count()*1e3);
  t0=std::chrono::steady_clock::now();
  { int fd=open("/tmp/w2.bin",O_WRONLY|O_CREAT|O_TRUNC,0644); (void)!ftruncate(fd, blob.size()); 
    std::vector<std::thread> th; for(int t=0;t<4;t++) th.emplace_back([&,t]{ size_t per=blob.size()/4; for(size_t i=0;i<per;i+=1<<20) (void)!pwrite(fd,blob.data()+t*per+i,1<<20,t*per+i);}); for(auto&x:th)x.join(); close(fd);}
  t1=std::chrono::steady_clock::now(); printf("4x pwrite() 140MB: %.1f ms\n", std::chrono::duration<double>(t1-t0).count()*1e3);
}
