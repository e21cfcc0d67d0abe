#!/usr/bin/perl -w
# chromosome3D_amd.pl — driver with the command line of the reference's chromosome3D.pl
# (-i|-if <IF matrix> -o <outdir> [-k 11] [-a 0.5] [-m 20] [-h]; reference :28-46, usage :2530-2557)
# whose solver leg is the MI355X library instead of CNS:
#
#   reference                                        here
#   IF2dist_new/dist2rr/carr2tbl (Perl, :87-89)      K1 kernel + host writers inside c3d_solve
#   build_extended (2 CNS runs, :102)                not needed (bead model, no pseudo-protein)
#   build_models: job.sh -> cns_solve < dgsa.inp     job.sh -> c3d_solve (C ABI of libc3d.so)
#   assess_dgsa (:106, :769-829)                     same ranking / table / renaming, below
#
# The FFI is the C ABI in include/c3d.h.  In-process route: the XS module bindings/perl (C3D::solve calls
# c3d_set_if_matrix / c3d_run / ... directly).  Fallback when the module is not built (or C3D_FORCE_CLI=1):
# job.sh + the c3d_solve executable, i.e. the reference's own process boundary.
use strict;
use warnings;
use Cwd 'abs_path';
use File::Basename;
use File::Copy;
use Getopt::Long;

my ($help, $dir_out, $file_if, $shape_only, $file_seq, $accepted);
my ($K, $ALPHA, $MODELS) = (11, 0.5, 20);        # chromosome3D.pl:18-21
my ($SEED, $DEVICE, $DISTRELAX) = (82364, 0, 0.5);  # :980, :74
GetOptions("h" => \$help, "o=s" => \$dir_out, "k=i" => \$K, "a=s" => \$ALPHA, "m=i" => \$MODELS,
           "i|if=s" => \$file_if, "seed=i" => \$SEED, "device=i" => \$DEVICE, "shape=s" => \$shape_only, "seq=s" => \$file_seq, "accepted" => \$accepted)
	or die "ERROR! Error in command line arguments!\n";
usage() if $help;
# residue names a model row may carry (the 20 standard amino acids; rows with any other name are dropped, reference :847)
my %AA = map { $_ => 1 } qw(ALA ASN CYS GLN HIS LEU MET PRO THR TYR ARG ASP GLU GLY ILE LYS PHE SER TRP VAL);
if (defined $shape_only) {   # test hook: only the output shaping of one solver PDB, in place (no GPU involved)
	open my $lg, ">>", ($dir_out // ".")."/model_info.log" or die $!;
	shape_pdb($shape_only, $lg);
	close $lg;
	exit 0;
}
usage("Input IF matrix not found!") if not defined $file_if;
usage("Output directory not defined!") if not defined $dir_out;
usage("Input IF file $file_if does not exist!") if not -f $file_if;

my $have_xs = 0;
if (not $ENV{C3D_FORCE_CLI}) {
	my $blib = abs_path(dirname(abs_path($0))."/../bindings/perl/blib");
	if (defined $blib and -f "$blib/C3D.pm") {
		unshift @INC, $blib;
		$have_xs = eval { require C3D; 1 } ? 1 : 0;
	}
}
my $solver = $ENV{C3D_SOLVE} || abs_path(dirname(abs_path($0))."/../chromosome3d_amd/_lib/c3d_solve");
die "ERROR! neither the C3D XS module nor c3d_solve ($solver) is built (python -c 'import __graft_entry__ as g; g.build()')\n" if not $have_xs and not -x $solver;

mkdir $dir_out or die "ERROR! Could not create output directory $dir_out!\n" if not -d $dir_out;
# C3D_TIMING=1: wall-clock of the driver's phases on stderr (Time::HiRes is core Perl)
my $t_mark;
sub tick { return if not $ENV{C3D_TIMING}; require Time::HiRes; my $t = Time::HiRes::time(); printf STDERR "[timing] %-28s %.3f s\n", $_[0], $t - ($t_mark // $^T); $t_mark = $t; }
tick("perl start, options, XS load");
print "Start Time : ".(localtime)." [$0]\n";
print "Input      : $file_if\nOutput Dir : $dir_out\nScaling(K) : $K\nAlpha      : $ALPHA\n";
print "Effective Conversion Equation is : D = $K * mean(IF^$ALPHA) / IF^$ALPHA\n";

my $ID = basename($file_if, ".txt");
# $ID names files, rides in glob() patterns and on job.sh's lines: letters, digits, '_', '.', '+', '-' only (a matrix called
# `a$(cmd).txt`, or one with quotes, blanks or glob characters, stops here, before anything is written)
die "ERROR! the matrix file's name may hold letters, digits, '_', '.', '+' and '-' only: '$ID'\n" unless $ID =~ /^[\w.+-]+$/;
# the reference wipes the whole output directory (`rm -f $dir_out/*`, :56); we only remove what a
# previous run of this driver left there
unlink glob("$dir_out/${ID}_*.pdb"), glob("$dir_out/${ID}a_*.pdb"), glob("$dir_out/iam.*");
unlink map { "$dir_out/$_" } ("$ID.dist", "$ID.rr", "contact.tbl", "job.sh", "job.log", "model_info.log", "contact_violation.txt");
copy($file_if, "$dir_out/$ID.txt") or die "ERROR! cannot copy $file_if: $!\n" if abs_path($file_if) ne (abs_path("$dir_out/$ID.txt") || "");
chdir $dir_out or die $!;

# <ID>.fasta (:92-98): one residue per bead.  Every bead is written as MET, as in the bundled output_models, so the
# sequence file that goes with the models is M x L (the reference cuts L letters out of a fixed pseudo-protein)
# -seq <fasta>: name the residues after the first L letters of that sequence instead, as a reference run does with its
# fixed 663-letter pseudo-protein (:93-94; chromosome3d_amd/data/refsequence.fasta holds it); beads beyond its end stay MET
my $sequence;
{
	my $L = first_line_fields("$ID.txt");
	$sequence = "M" x $L;
	if (defined $file_seq) {
		open my $sq, "<", $file_seq or die "ERROR! cannot read $file_seq: $!\n";
		my $letters = join "", map { s/\s+//gr } grep { !/^>/ } <$sq>;
		close $sq;
		# one-letter residue codes only: the letters end up in <ID>.fasta and name residues, nothing else may ride along
		# (a '*' stop codon, ';' or '$(...)' must never reach a shell or a file name)
		die "ERROR! $file_seq: residue codes must be letters A-Z (found '$1')\n" if $letters =~ /([^A-Za-z])/;
		$letters = uc $letters;
		$sequence = substr($letters, 0, $L);
		$sequence .= "M" x ($L - length $sequence);
	}
	open my $fa, ">", "$ID.fasta" or die $!;
	print $fa ">$ID\n", $sequence, "\n";
	close $fa;
}

# (B) build models
my $restraints;
if ($have_xs) {
	# in-process FFI: Perl -> XS -> C ABI -> HIP kernels
	print "(B) Build models using libc3d (MI355X) through the C3D XS binding..\n";
	system("touch iam.running");
	C3D::set_sequence(defined $file_seq ? $sequence : "");
	my $r = eval { C3D::solve("$ID.txt", ".", $ID, $MODELS, $K, $ALPHA + 0, $SEED, $DEVICE, 0) };
	if (not defined $r) {
		rename "iam.running", "iam.failed";
		die "ERROR! Something went wrong inside libc3d: $@";
	}
	unlink "iam.running";
	if ($accepted) { copy("${ID}_$_.pdb", "${ID}a_$_.pdb") or die "ERROR! cannot write ${ID}a_$_.pdb: $!\n" foreach (1 .. $MODELS); }   # (-accepted: see below)
	$restraints = $r->{restraints};
	open my $jl, ">", "job.log" or die $!;
	printf $jl "C3D XS binding: %d beads, %d restraints, %d models, %d SA steps in %.1f ms on device %d\n", $r->{n}, $r->{restraints}, $MODELS, $r->{steps}, $r->{ms}, $DEVICE;
	close $jl;
}
else {
# the process boundary of the reference, same sentinel protocol (:258-288)
open my $job, ">", "job.sh" or die $!;
print $job "#!/bin/bash\necho \"starting c3d_solve..\"\ntouch iam.running\n";
# every string argument single-quoted for the shell; the residue names travel as the file written above, never as text on the line
print $job shq($solver)." --if ".shq("$ID.txt")." --out . --id ".shq($ID)." -k ".($K+0)." -a ".($ALPHA+0)." -m ".int($MODELS)." --seed ".int($SEED)." --device ".int($DEVICE).(defined $file_seq ? " --seq ".shq("\@$ID.fasta") : "").($accepted ? " --accepted" : "")."\n";
print $job "if [ -f ".shq("${ID}_".int($MODELS).".pdb")." ]; then\n   rm -f iam.running\n   echo \"trial structures written.\"\n   exit\nfi\n";
print $job "if [ -f ".shq("${ID}a_".int($MODELS).".pdb")." ]; then\n   rm -f iam.running\n   echo \"accepted structures written.\"\n   exit\nfi\n";   # (:272-277)
print $job "echo \"ERROR! Final structures not found!\"\nmv iam.running iam.failed 2>/dev/null || touch iam.failed\n";
close $job;
chmod 0755, "job.sh";
print "(B) Build models using libc3d (MI355X)..\nStarting job [$dir_out/job.sh > job.log]\n";
system("./job.sh > job.log 2>&1");
die "ERROR! Something went wrong while running c3d_solve! Check job.log!\n".`tail -n 5 job.log` if -f "iam.failed" or not (-f "${ID}_${MODELS}.pdb" or -f "${ID}a_${MODELS}.pdb");
($restraints) = `cat job.log` =~ /Restraints : (\d+)/;
}
tick("(B) build models");
print "L          : ".first_line_fields("$ID.txt")."\n";
print "Restraints : ".($restraints // "?")." lines in tbl file\n";

# (C) assess models: rank by int(REMARK noe) ascending (:796-802), table (:804-810), top 5 (:822-828)
print "(C) Assess models..\n";
# remove the "trial" structure of a corresponding "accepted" structure because they are the same (:790-795).  libc3d writes accepted twins
# only when asked (-accepted: CNS's acceptance thresholds are defined on covalent geometry a bead model does not have, so every model is
# accepted then); the ranking below then runs on the <ID>a_<k>.pdb files, as the reference's does for structures CNS accepted
for my $i (0 .. 1000) {
	next if not -f "${ID}a_$i.pdb";
	print "\ndeleting ${ID}_$i.pdb because ${ID}a_$i.pdb exists!";
	unlink "./${ID}_$i.pdb";
}
my %e_noe;
foreach my $pdb (glob("./${ID}_*.pdb"), glob("./${ID}a_*.pdb")) {
	next if $pdb =~ /_model\d+\.pdb$/;
	open my $fh, "<", $pdb or die $!;
	my $v;
	while (<$fh>) { if (/^REMARK noe/) { (my $t = $_) =~ s/\s+//g; $v = (split /=/, $t)[1]; } }
	close $fh;
	die "ERROR! $pdb has no REMARK noe line!\n" if not defined $v;
	$e_noe{$pdb} = int($v);
}
open my $log, ">>", "model_info.log" or die $!;
print "\nNOE_SATISFIED(+-${DISTRELAX}A)  SUM_OF_DEVIATIONS>= 0.2  PDB\n";
# The satisfaction numbers and the rows of contact_violation.txt (count_satisfied_tbl_rows :447-485, sum_noe_dev :581-600) come from the
# library (c3d_write_violations: the same arithmetic on the PDB's %8.3f coordinates, the same row format) — through the XS binding or, without
# it, through c3d_score --assess; 20 models x 101 426 rows take 0.3 s there and 4 s in the Perl loop below, which stays as the last resort.
my $scorer = dirname($solver)."/c3d_score";
my $by_lib = ($have_xs and defined &C3D::assess) ? 1 : (-x $scorer ? 2 : 0);
my @ordered = sort { $e_noe{$b} <=> $e_noe{$a} || $a cmp $b } keys %e_noe;
if ($by_lib == 1) {
	my @r = C3D::assess("contact.tbl", $DISTRELAX + 0, "contact_violation.txt", @ordered);      # (count, total, sum_dev) per model
	foreach my $pdb (@ordered) {
		my ($count, $total, $sum_dev) = splice(@r, 0, 3);
		printf "%-9s             %-9s                %-25s\n", "$count/$total", (sprintf "%.2f", $sum_dev), basename($pdb, ".pdb");
	}
}
elsif ($by_lib == 2) {
	my @out = `@{[shq($scorer)]} --assess contact.tbl $DISTRELAX contact_violation.txt @{[join " ", map { shq($_) } @ordered]}`;
	die "ERROR! c3d_score --assess failed\n" if $? != 0 or @out != @ordered;
	foreach my $k (0 .. $#ordered) {
		die "ERROR! c3d_score --assess: unexpected row $out[$k]\n" if $out[$k] !~ /^(\d+\/\d+)\t(-?[\d.]+)\t/;
		printf "%-9s             %-9s                %-25s\n", $1, $2, basename($ordered[$k], ".pdb");
	}
}
my @tbl = $by_lib ? () : read_tbl("contact.tbl");
foreach my $pdb ($by_lib ? () : @ordered) {
	my %xyz = read_ca($pdb);
	my ($count, $total, $sum_dev) = (0, 0, 0.0);
	my (@viol, @ok);   # rows of contact_violation.txt (reference :475-483: violated rows first)
	foreach my $r (@tbl) {
		my ($i, $j, $t) = @$r;
		my $d = sprintf "%.3f", sqrt(($xyz{$i}[0]-$xyz{$j}[0])**2 + ($xyz{$i}[1]-$xyz{$j}[1])**2 + ($xyz{$i}[2]-$xyz{$j}[2])**2);
		my ($flag, $deviation) = (1, $d - $t);
		if ($d < $t + $DISTRELAX) { $count++; $flag = 0; $deviation = 0.0; }
		if ($d < $t - $DISTRELAX) { $count--; $flag = 1; $deviation = -($t - $d); }
		$sum_dev += $d - $t if $d > $t + 0.2;
		$sum_dev += $t - $d if $d < $t - 0.2;
		$total++;
		my $row = sprintf "%3s\t%.2f\t%.2f # assign45  resid %3d and name ca   resid %3d and name ca  %.2f 0.00 0.00", $flag, $deviation, $d, $i, $j, $t;
		if ($flag) { push @viol, $row } else { push @ok, $row }
	}
	open my $vf, ">>", "contact_violation.txt" or die $!;
	print $vf "#NOE violation check; $pdb against contact.tbl\n#violation-flag, deviation, actual-measurement, Input-NOE-restraint\n";
	print $vf "$_\n" foreach (@viol, @ok);
	close $vf;
	printf "%-9s             %-9s                %-25s\n", "$count/$total", (sprintf "%.2f", $sum_dev), basename($pdb, ".pdb");
}
tick("(C) satisfaction + violation table");
print "\n";
print "removing non-CA ATOM rows and backing up REMARK rows..\n";
shape_pdb($_, $log) foreach (sort { $e_noe{$b} <=> $e_noe{$a} || $a cmp $b } keys %e_noe);
close $log;
print "\n";
my $rank = 1;
foreach my $pdb (sort { $e_noe{$a} <=> $e_noe{$b} || $a cmp $b } keys %e_noe) {
	print "model$rank.pdb <= $pdb\n";
	rename $pdb, "${ID}_model$rank.pdb" or die $!;
	last if ++$rank > 5;
}
tick("shaping, top five");
print "\nFinished [$0]: ".(localtime)."\n";

# Output shaping of one model, in place — what the reference's assess_dgsa leaves behind (:813-820): REMARK rows go to
# model_info.log behind the file name; ATOM rows that mention CA stay, atoms and residues renumbered from 1,
# chain id blanked; an empty line (the reference writes END there and then deletes the word); CONECT i i+1; END.
sub shape_pdb {
	my ($pdb, $logfh) = @_;
	open my $in, "<", $pdb or die "ERROR! cannot read $pdb: $!\n";
	my @rows = <$in>;
	close $in;
	print $logfh $pdb;   # the reference's log has the name glued to the first REMARK row (print2line, :870)
	my ($atoms, $res, $prev, $nca, @out) = (0, 0, "XX", 0);
	foreach my $r (@rows) {
		print $logfh $r if $r =~ /^REMARK/;
		next if $r !~ /^ATOM/ or $r !~ /CA/;
		my ($alt, $rname, $rnum, $aname) = map { (my $t = length($r) > $_->[0] ? substr($r, $_->[0], $_->[1]) : "") =~ s/\s+//g; $t } ([16, 1], [17, 3], [22, 5], [12, 4]);
		next if not ($alt eq "" or $alt eq "A");
		next if not exists $AA{$rname};
		if ($rnum ne $prev) { $prev = $rnum; $res++; }
		$atoms++;
		$nca++ if $aname eq "CA";
		push @out, substr($r, 0, 6).sprintf("%5s", $atoms).substr($r, 11, 5)." ".substr($r, 17, 3)."  ".sprintf("%4s", $res)." ".substr($r, 27);
	}
	die "ERROR! $pdb has less than 1 residue!\n" if $nca < 1;
	open my $o, ">", $pdb or die "ERROR! cannot write $pdb: $!\n";
	print $o @out, "\n";
	printf $o "CONECT%5s%5s\n", $_, $_ + 1 foreach (1 .. $nca - 1);
	print $o "END\n";
	close $o;
}
# single-quote a string for /bin/sh
sub shq { my $s = shift; $s =~ s/'/'\\''/g; return "'$s'"; }
sub first_line_fields { open my $f, "<", shift or die $!; my $l = <$f>; close $f; $l =~ s/^\s+//; my @t = split /\s+/, $l; return scalar @t; }
sub read_tbl {
	my @rows;
	open my $f, "<", shift or die $!;
	while (<$f>) { tr/()/  /; my @c = split; next if not @c; die "bad tbl row: $_" if $c[0] !~ /^assign/; push @rows, [$c[2], $c[7], $c[11]]; }
	close $f;
	return @rows;
}
sub read_ca {
	my %xyz;
	open my $f, "<", shift or die $!;
	while (<$f>) { next if !/^ATOM/; (my $an = substr($_, 12, 4)) =~ s/\s+//g; next if $an ne "CA";
		(my $rn = substr($_, 22, 5)) =~ s/\s+//g; $xyz{$rn} = [substr($_, 30, 8) + 0, substr($_, 38, 8) + 0, substr($_, 46, 8) + 0]; }
	close $f;
	return %xyz;
}
sub usage {
	my $msg = shift;
	print "\nERROR! $msg\n" if defined $msg;
	print <<"EOU";

PARAM        DESCRIPTION
-i | -if  :  Input IF matrix (N x N, whitespace separated)
-o        :  Output directory
-k        :  Scaling constant K (default 11)
-a        :  Alpha for the IF -> distance conversion (default 0.5)
-m        :  Number of models to generate (default 20)
--seed    :  RNG seed (default 82364)   --device : GPU index (default 0)
--seq     :  FASTA file naming the residues (the reference's pseudo-protein: chromosome3d_amd/data/refsequence.fasta; default: all MET)
--accepted:  also write the accepted twin <ID>a_<k>.pdb of every model, as CNS does for structures it accepts; the trial twins are then dropped and
             the accepted files ranked, as the reference's assess_dgsa does (default: trial files only)
Example: $0 -i ./input/chr22_1mb_matrix.txt -o ./output/chr22_1mb
EOU
	exit(defined $msg ? 1 : 0);
}
