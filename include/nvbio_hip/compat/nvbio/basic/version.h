// compat/nvbio/basic/version.h -- the version macros applications print (nvbio/basic/version.h): the reference release whose
// interface this layer follows.
#pragma once
#define NVBIO_VERSION           100150
#define NVBIO_MAJOR_VERSION     (NVBIO_VERSION / 100000)
#define NVBIO_MINOR_VERSION     (NVBIO_VERSION / 100 % 1000)
#define NVBIO_SUBMINOR_VERSION  (NVBIO_VERSION % 100)
#define NVBIO_VERSION_STRING    "1.1.50"
