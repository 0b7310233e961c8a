//! g1s-ref-diff SOURCE.y4m DENOISED.y4m OUT.tbl
//!
//! The reference's `diff` arithmetic without its ffmpeg front end: av1_grain::DiffGenerator (the crate
//! grav1synth 0.4.x pins) driven exactly as grav1synth's src/main.rs:414-529 drives it -- new(fps, source bit
//! depth, denoised bit depth), diff_frame per frame pair in order, finish(), then the table in GRAV1SYNTH'S OWN text
//! layout (its src/main.rs:525-529 and write_film_grain_segment, :631-696 -- "filmgrn1", two spaces after "sY n", one after
//! "sCb n" / "sCr n" -- restated below, not av1_grain::write_grain_table: a byte that differs from the committed tables is
//! then arithmetic, not whitespace).  The Y4M
//! reader below is this harness's own (8 / 10 / 12-bit, 4:2:0 / 4:2:2 / 4:4:4 / mono), so the binary needs
//! no video libraries.  Test infrastructure: it produces golden tables for tests/test_reference_pin.py and is
//! never linked into or called by the product.
use std::io::{BufRead, BufReader, BufWriter, Read, Write};

use anyhow::{anyhow, bail, Result};
use av1_grain::{DiffGenerator, GrainTableSegment};
use num_rational::Rational64;
use v_frame::{frame::Frame, pixel::{ChromaSampling, Pixel}};

struct Y4m { r: BufReader<std::fs::File>, w: usize, h: usize, fps: Rational64, bd: usize, cs: ChromaSampling }

fn open(path: &str) -> Result<Y4m> {
    let mut r = BufReader::new(std::fs::File::open(path)?);
    let mut line = String::new();
    r.read_line(&mut line)?;
    let mut it = line.trim_end().split(' ');
    if it.next() != Some("YUV4MPEG2") { bail!("{path}: not a YUV4MPEG2 file"); }
    let (mut w, mut h, mut fps, mut bd, mut cs) = (0, 0, Rational64::new(25, 1), 8, ChromaSampling::Cs420);
    for tag in it {
        let (k, v) = tag.split_at(1);
        match k {
            "W" => w = v.parse()?,
            "H" => h = v.parse()?,
            "F" => { let (n, d) = v.split_once(':').ok_or_else(|| anyhow!("bad F tag"))?; fps = Rational64::new(n.parse()?, d.parse()?); }
            "C" => {
                cs = if v.starts_with("420") { ChromaSampling::Cs420 } else if v.starts_with("422") { ChromaSampling::Cs422 }
                     else if v.starts_with("444") { ChromaSampling::Cs444 } else if v.starts_with("mono") { ChromaSampling::Cs400 }
                     else { bail!("unsupported colour space {v}") };
                bd = if v.contains("p10") { 10 } else if v.contains("p12") { 12 } else { 8 };
            }
            _ => {}
        }
    }
    Ok(Y4m { r, w, h, fps, bd, cs })
}

fn next<T: Pixel>(y: &mut Y4m) -> Result<Option<Frame<T>>> {
    let mut line = String::new();
    if y.r.read_line(&mut line)? == 0 { return Ok(None); }
    if !line.starts_with("FRAME") { bail!("expected FRAME, found {line:?}"); }
    let mut f: Frame<T> = Frame::new_with_padding(y.w, y.h, y.cs, 0);
    let bytes = if y.bd > 8 { 2 } else { 1 };
    let planes = if y.cs == ChromaSampling::Cs400 { 1 } else { 3 };
    for p in 0..planes {
        let (pw, ph) = (f.planes[p].cfg.width, f.planes[p].cfg.height);
        let mut buf = vec![0u8; pw * ph * bytes];
        y.r.read_exact(&mut buf)?;
        f.planes[p].copy_from_raw_u8(&buf, pw * bytes, bytes);
    }
    Ok(Some(f))
}

fn run<T: Pixel, U: Pixel>(mut s: Y4m, mut d: Y4m, out: &str) -> Result<()> {
    let mut differ = DiffGenerator::new(s.fps, s.bd, d.bd);
    let mut frames = 0usize;
    loop {
        match (next::<T>(&mut s)?, next::<U>(&mut d)?) {
            (Some(a), Some(b)) => differ.diff_frame(&a, &b)?,
            (None, None) => break,
            _ => { eprintln!("Videos did not have equal frame counts."); break; }
        }
        frames += 1;
    }
    write_tbl(out, &differ.finish())?;
    eprintln!("Computed diff for {frames} frames");
    Ok(())
}

/// The `.tbl` text as grav1synth writes it after `diff` (reference src/main.rs:525-529, 631-696): the fields of
/// av1_grain::GrainTableSegment in the order of its `From` impl (src/parser/grain.rs:108-133).
fn write_tbl(path: &str, segments: &[GrainTableSegment]) -> Result<()> {
    let mut o = BufWriter::new(std::fs::File::create(path)?);
    writeln!(o, "filmgrn1")?;
    for s in segments {
        writeln!(o, "E {} {} 1 {} 1", s.start_time, s.end_time, s.random_seed)?;
        writeln!(
            o,
            "\tp {} {} {} {} {} {} {} {} {} {} {} {}",
            s.ar_coeff_lag, s.ar_coeff_shift, s.grain_scale_shift, s.scaling_shift,
            u8::from(s.chroma_scaling_from_luma), u8::from(s.overlap_flag),
            s.cb_mult, s.cb_luma_mult, s.cb_offset, s.cr_mult, s.cr_luma_mult, s.cr_offset
        )?;
        write!(o, "\tsY {} ", s.scaling_points_y.len())?; // (a space here AND one before every point: two after the count)
        for p in &s.scaling_points_y { write!(o, " {} {}", p[0], p[1])?; }
        writeln!(o)?;
        write!(o, "\tsCb {}", s.scaling_points_cb.len())?;
        for p in &s.scaling_points_cb { write!(o, " {} {}", p[0], p[1])?; }
        writeln!(o)?;
        write!(o, "\tsCr {}", s.scaling_points_cr.len())?;
        for p in &s.scaling_points_cr { write!(o, " {} {}", p[0], p[1])?; }
        writeln!(o)?;
        write!(o, "\tcY")?;
        for c in &s.ar_coeffs_y { write!(o, " {}", *c)?; }
        writeln!(o)?;
        write!(o, "\tcCb")?;
        for c in &s.ar_coeffs_cb { write!(o, " {}", *c)?; }
        writeln!(o)?;
        write!(o, "\tcCr")?;
        for c in &s.ar_coeffs_cr { write!(o, " {}", *c)?; }
        writeln!(o)?;
    }
    o.flush()?;
    Ok(())
}

fn main() -> Result<()> {
    let a: Vec<String> = std::env::args().collect();
    if a.len() != 4 { bail!("usage: g1s-ref-diff SOURCE.y4m DENOISED.y4m OUT.tbl"); }
    let (s, d) = (open(&a[1])?, open(&a[2])?);
    match (s.bd > 8, d.bd > 8) {
        (false, false) => run::<u8, u8>(s, d, &a[3]),
        (false, true) => run::<u8, u16>(s, d, &a[3]),
        (true, false) => run::<u16, u8>(s, d, &a[3]),
        (true, true) => run::<u16, u16>(s, d, &a[3]),
    }
}
