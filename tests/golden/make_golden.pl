#!/usr/bin/perl -w
# Golden-vector generator.  TEST INFRASTRUCTURE, runs only in the build container.
#
# Loads the *reference's own* Perl subs from /root/reference/chromosome3D.pl at run
# time (text is read, eval'ed in this process and never written anywhere), runs them on
# the bundled inputs and writes the fixtures under tests/golden/.  No reference source
# text lives in this file or in the fixtures: the fixtures are inputs and expected outputs.
#
#   perl tests/golden/make_golden.pl [/root/reference] [tests/golden]
#
# Subs exercised (reference file:line):
#   IF2dist_new 110-162, calc_len_IF 164-179, dist2rr 181-206, carr2tbl 340-362,
#   count_satisfied_tbl_rows 447-485, sum_noe_dev 581-600, clash_count 693-714
#   (+ the helpers they call: system_cmd, print2file, print2line, count_lines,
#    ssnoe_tbl_min_pdb_dist, xyz_pdb, pdb2rnum_rname, parse_pdb_row, calc_dist).
use strict;
use warnings;
use Carp;
use Cwd 'abs_path';
use File::Basename;
use File::Temp qw(tempdir);
use File::Copy;
use Digest::MD5;
use Scalar::Util qw(looks_like_number);

my $ref = shift || "/root/reference";
my $out = abs_path(shift || dirname(abs_path($0)));
my $script = "$ref/chromosome3D.pl";
die "reference script $script not found (this generator only runs where the reference is mounted)\n" if not -f $script;

# ---- globals the reference subs close over (chromosome3D.pl:17-21, 64-78) -------------
our ($L, $ALPHA, $KSCALING, $min_sep, $DISTRELAX, %AA3TO1, %AA1TO3);
$ALPHA = 0.5; $KSCALING = 11; $min_sep = 5; $DISTRELAX = 0.5;
%AA3TO1 = qw(ALA A ASN N CYS C GLN Q HIS H LEU L MET M PRO P THR T TYR Y ARG R ASP D GLU E GLY G ILE I LYS K PHE F SER S TRP W VAL V);
%AA1TO3 = reverse %AA3TO1;

# ---- pull the named sub bodies out of the reference and eval them ----------------------
my @wanted = qw(IF2dist_new calc_len_IF dist2rr carr2tbl system_cmd count_lines print2file
                print2line count_satisfied_tbl_rows ssnoe_tbl_min_pdb_dist sum_noe_dev
                pdb2rnum_rname xyz_pdb parse_pdb_row clash_count calc_dist);
my %want = map { $_ => 1 } @wanted;
open my $fh, "<", $script or die $!;
my ($cur, %body);
while (my $line = <$fh>) {
	if (not defined $cur and $line =~ /^sub\s+(\w+)\s*\{/) { $cur = $1; $body{$cur} = ""; }
	if (defined $cur) {
		$body{$cur} .= $line;
		undef $cur if $line =~ /^\}/;
	}
}
close $fh;
my $code = "no strict 'vars'; no warnings;\n";
foreach (@wanted) { die "sub $_ not found in reference\n" if not defined $body{$_}; $code .= $body{$_}; }
$code =~ s/\bmy \$L\b/my \$L_local/g if 0;    # (the subs use the file-global $L; we supply it)
eval $code; die "eval of reference subs failed: $@" if $@;

sub md5_of { my $f = shift; open my $h, "<", $f or die $!; binmode $h; my $d = Digest::MD5->new->addfile($h)->hexdigest; close $h; return $d; }

# ---- front half on a set of bundled inputs ----------------------------------------------
my @full  = qw(chr21_1mb chr22_1mb);                          # committed in full (small)
my @sums  = qw(chr1_500kb chr19_500kb chr13_1mb chr20_1mb chr4_1mb);  # md5 + counts only
my %summary;
my $here = abs_path(".");
foreach my $id (@full, @sums) {
	my $tmp = tempdir(CLEANUP => 1);
	copy("$ref/input/${id}_matrix.txt", "$tmp/$id.txt") or die $!;
	chdir $tmp or die $!;
	$L = calc_len_IF("$id.txt");
	IF2dist_new("$id.txt", "$id.dist", $KSCALING);
	dist2rr("$id.dist", "$id.rr");
	carr2tbl("$id.rr", "contact.tbl");
	my $lines = count_lines("contact.tbl");
	$summary{$id} = { n => $L, restraints => $lines, md5_tbl => md5_of("contact.tbl"),
	                  md5_dist => md5_of("$id.dist"), md5_rr => md5_of("$id.rr") };
	if (grep { $_ eq $id } @full) {
		copy("$id.dist", "$out/$id.dist") or die $!;
		copy("$id.rr", "$out/$id.rr") or die $!;
		copy("contact.tbl", "$out/$id.contact.tbl") or die $!;
	}
	# assessment known answers against the bundled model of this chromosome
	my @models = glob("$ref/output_models/${id}_rank*_a11.pdb");
	if (@models) {
		my $pdb = $models[0];
		copy($pdb, "$tmp/model.pdb") or die $!;
		my $sat = count_satisfied_tbl_rows("model.pdb", "contact.tbl", "noe");
		my $dev = sum_noe_dev("model.pdb", "contact.tbl");
		my $clash = clash_count("model.pdb", 3.5);
		$summary{$id}{model} = basename($pdb);
		$summary{$id}{satisfied} = $sat;
		$summary{$id}{sum_dev} = $dev;
		$summary{$id}{clash_3p5} = $clash;
		if ($id eq "chr21_1mb") { copy("contact_violation.txt", "$out/$id.contact_violation.txt") or die $!; }
	}
	chdir $here;
}

# ---- write the summary as JSON (hand-rolled: no JSON module assumed) ---------------------
open my $js, ">", "$out/front_half_golden.json" or die $!;
print $js "{\n";
my @ids = sort keys %summary;
for (my $k = 0; $k <= $#ids; $k++) {
	my $id = $ids[$k]; my $s = $summary{$id};
	my @kv;
	foreach my $key (sort keys %$s) {
		my $v = $s->{$key};
		push @kv, ($v =~ /^-?\d+(\.\d+)?$/ and $key !~ /md5/) ? "\"$key\": $v" : "\"$key\": \"$v\"";
	}
	print $js "  \"$id\": {".join(", ", @kv)."}".($k < $#ids ? "," : "")."\n";
}
print $js "}\n";
close $js;
print "wrote $out/front_half_golden.json\n";
