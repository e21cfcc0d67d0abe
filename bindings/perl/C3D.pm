package C3D;
# Perl side of the XS binding of libc3d (include/c3d.h).  Build: make -C bindings/perl
use strict;
use warnings;
our $VERSION = '0.1';
require XSLoader;
XSLoader::load('C3D', $VERSION);
1;
