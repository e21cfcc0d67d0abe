#!/usr/bin/perl -w
# Output-side golden generator.  TEST INFRASTRUCTURE, runs only in the build container.
#
# Takes solver-output PDBs written by OUR writer (c3d_write_pdb, through tests/golden/make_output_golden.py) and
# runs the *reference's own* post-processing on them: the subs are read from /root/reference/chromosome3D.pl at run
# time, eval'ed in this process and never written anywhere.  What assess_dgsa does to every model
# (chromosome3D.pl:796-828): get_cns_energy :602-618, count_satisfied_tbl_rows :447-485, sum_noe_dev :581-600,
# filter_nonCA :864-880, reindex_chain :831-862, `sed -i "s/END//g"` :818, add_connect_rows :208-215.
# Fixtures written: <id>_final.pdb (the file the reference leaves behind), <id>_model_info.log,
# output_side_golden.json (int(noe), satisfied, sum_dev per input).
#
#   perl tests/golden/make_output_golden.pl /root/reference tests/golden/output_side
use strict;
use warnings;
use Carp;
use Cwd 'abs_path';
use File::Basename;
use File::Temp qw(tempdir);
use File::Copy;
use Scalar::Util qw(looks_like_number);

my $ref = shift || "/root/reference";
my $dir = abs_path(shift || dirname(abs_path($0))."/output_side");
my $script = "$ref/chromosome3D.pl";
die "reference script $script not found (this generator only runs where the reference is mounted)\n" if not -f $script;

our ($L, $DISTRELAX, %AA3TO1, %AA1TO3, $model_info_log);
$DISTRELAX = 0.5;
%AA3TO1 = qw(ALA A ASN N CYS C GLN Q HIS H LEU L MET M PRO P THR T TYR Y ARG R ASP D GLU E GLY G ILE I LYS K PHE F SER S TRP W VAL V);
%AA1TO3 = reverse %AA3TO1;

my @wanted = qw(get_cns_energy filter_nonCA reindex_chain add_connect_rows seq_chain print2file print2line system_cmd
                count_lines count_satisfied_tbl_rows ssnoe_tbl_min_pdb_dist sum_noe_dev pdb2rnum_rname xyz_pdb
                parse_pdb_row calc_dist);
open my $fh, "<", $script or die $!;
my ($cur, %body);
while (my $line = <$fh>) {
	if (not defined $cur and $line =~ /^sub\s+(\w+)\s*\{/) { $cur = $1; $body{$cur} = ""; }
	if (defined $cur) { $body{$cur} .= $line; undef $cur if $line =~ /^\}/; }
}
close $fh;
my $code = "no strict 'vars'; no warnings;\n";
foreach (@wanted) { die "sub $_ not found in reference\n" if not defined $body{$_}; $code .= $body{$_}; }
eval $code; die "eval of reference subs failed: $@" if $@;

my %summary;
my $here = abs_path(".");
foreach my $in (sort glob("$dir/*_solver_out.pdb")) {
	(my $id = basename($in)) =~ s/_solver_out\.pdb$//;
	my $tbl = "$dir/../$id.contact.tbl";
	die "no contact.tbl fixture for $id\n" if not -f $tbl;
	my $tmp = tempdir(CLEANUP => 1);
	copy($in, "$tmp/${id}_1.pdb") or die $!;
	copy($tbl, "$tmp/contact.tbl") or die $!;
	chdir $tmp or die $!;
	$model_info_log = "model_info.log";
	my $pdb = "./${id}_1.pdb";
	# the order of assess_dgsa: energy and table on the solver's file, then the shaping
	my $e = get_cns_energy($pdb, "noe");
	my $n1 = count_satisfied_tbl_rows($pdb, "contact.tbl", "noe");
	my $s1 = sum_noe_dev($pdb, "contact.tbl");
	system_cmd("rm -f ca_filtered.pdb");
	filter_nonCA($pdb, "ca_filtered.pdb", $model_info_log);
	reindex_chain("ca_filtered.pdb", 1, $pdb);
	system_cmd("rm -f ca_filtered.pdb");
	system_cmd("sed -i \"s/END//g\" $pdb");
	add_connect_rows($pdb);
	copy($pdb, "$dir/${id}_final.pdb") or die $!;
	copy($model_info_log, "$dir/${id}_model_info.log") or die $!;
	$summary{$id} = { noe_int => $e, satisfied => $n1, sum_dev => $s1, table_row => sprintf("%-9s             %-9s                %-25s", $n1, $s1, basename($pdb, ".pdb")) };
	chdir $here;
}
open my $js, ">", "$dir/output_side_golden.json" or die $!;
print $js "{\n";
my @ids = sort keys %summary;
for (my $k = 0; $k <= $#ids; $k++) {
	my $s = $summary{$ids[$k]};
	my @kv = map { my $v = $s->{$_}; ($v =~ /^-?\d+(\.\d+)?$/) ? "\"$_\": $v" : "\"$_\": \"$v\"" } sort keys %$s;
	print $js "  \"$ids[$k]\": {".join(", ", @kv)."}".($k < $#ids ? "," : "")."\n";
}
print $js "}\n";
close $js;
print "wrote $dir/output_side_golden.json (".scalar(@ids)." models)\n";
