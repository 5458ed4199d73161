//! `ibu-hip`: the public API of the `ibu` crate (reference `src/lib.rs:178-181`) re-created on
//! top of `libibu_hip.so`.  SOURCE ONLY — this image has no Rust toolchain, so this file has
//! never been compiled; it documents the binding a maintainer would add.
//!
//! Same names, argument meaning and error behaviour as the reference:
//! `Header, Record, HEADER_SIZE, MAGIC, RECORD_SIZE, VERSION, IbuError, IntoIbuError, Result,
//! load_to_vec, MmapReader, Reader, Writer, ParallelProcessor, ParallelReader`, plus
//! `device::{Context, DeviceBuf}` for the HIP kernels.
pub mod ffi;

use std::ffi::{c_void, CStr, CString};
use std::io::{Read, Write};
use std::marker::PhantomData;
use std::path::Path;

pub use ffi::ibu_header_t as Header;
pub use ffi::ibu_record_t as Record;

pub const MAGIC: u32 = 0x21554249;
pub const VERSION: u32 = 2;
pub const HEADER_SIZE: usize = 32;
pub const RECORD_SIZE: usize = 24;

// ---- errors (reference src/error.rs:56-128) ---------------------------------------------
#[derive(thiserror::Error, Debug)]
pub enum IbuError {
    #[error("I/O error")]
    Io(#[from] std::io::Error),
    #[error("Niffler error: {0}")]
    Niffler(String),
    #[error("Invalid magic number, expected ({expected:#x}), found ({actual:#x})")]
    InvalidMagicNumber { expected: u32, actual: u32 },
    #[error("Truncated record at position {pos}")]
    TruncatedRecord { pos: usize },
    #[error("Invalid version found, expected ({expected}), found ({actual})")]
    InvalidVersion { expected: u32, actual: u32 },
    #[error("Invalid barcode length: {0} (must be 1-32)")]
    InvalidBarcodeLength(u32),
    #[error("Invalid UMI length: {0} (must be 1-32)")]
    InvalidUmiLength(u32),
    #[error("Invalid map size - not a multiple of record size")]
    InvalidMapSize,
    #[error("Invalid index ({idx}) - Must be less than {max}")]
    InvalidIndex { idx: usize, max: usize },
    #[error("Processing error: {0}")]
    Process(Box<dyn std::error::Error + Send + Sync>),
    /// codec: a byte outside ACGTacgt (the reference leaves this to `bitnuc`)
    #[error("Invalid base: {n_bad} record(s), first at {first_bad}")]
    InvalidBase { first_bad: u64, n_bad: u64 },
    #[error("device/runtime error {code}: {message}")]
    Device { code: i32, message: String },
}
pub type Result<T> = std::result::Result<T, IbuError>;

pub trait IntoIbuError {
    fn into_ibu_error(self) -> IbuError;
}
impl<E: std::error::Error + Send + Sync + 'static> IntoIbuError for E {
    fn into_ibu_error(self) -> IbuError {
        IbuError::Process(self.into())
    }
}

fn check(rc: i32) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let mut d = std::mem::MaybeUninit::<ffi::ibu_error_detail_t>::zeroed();
    let d = unsafe {
        ffi::ibu_last_error(d.as_mut_ptr());
        d.assume_init()
    };
    let msg = unsafe { CStr::from_ptr(d.message.as_ptr()) }.to_string_lossy().into_owned();
    Err(match rc {
        1 => IbuError::Io(if d.os_errno != 0 {
            std::io::Error::from_raw_os_error(d.os_errno)
        } else {
            std::io::Error::new(std::io::ErrorKind::UnexpectedEof, msg)
        }),
        2 => IbuError::Niffler(msg),
        3 => IbuError::InvalidMagicNumber { expected: d.a as u32, actual: d.b as u32 },
        4 => IbuError::TruncatedRecord { pos: d.a as usize },
        5 => IbuError::InvalidVersion { expected: d.a as u32, actual: d.b as u32 },
        6 => IbuError::InvalidBarcodeLength(d.a as u32),
        7 => IbuError::InvalidUmiLength(d.a as u32),
        8 => IbuError::InvalidMapSize,
        9 => IbuError::InvalidIndex { idx: d.a as usize, max: d.b as usize },
        10 => IbuError::Process(msg.into()),
        11 => IbuError::InvalidBase { first_bad: d.a, n_bad: d.b },
        code => IbuError::Device { code, message: msg },
    })
}

// ---- Header / Record inherent methods (header.rs:84-228, record.rs:87-132) -----------------
impl Header {
    pub fn new(bc_len: u32, umi_len: u32) -> Self {
        let mut h = bytemuck::Zeroable::zeroed();
        unsafe { ffi::ibu_header_init(&mut h, bc_len, umi_len) };
        h
    }
    pub fn set_sorted(&mut self) {
        unsafe { ffi::ibu_header_set_sorted(self) }
    }
    pub fn sorted(&self) -> bool {
        unsafe { ffi::ibu_header_sorted(self) != 0 }
    }
    pub fn validate(&self) -> Result<()> {
        check(unsafe { ffi::ibu_header_validate(self) })
    }
    pub fn as_bytes(&self) -> &[u8] {
        bytemuck::bytes_of(self)
    }
    pub fn from_bytes(bytes: &[u8]) -> Self {
        *bytemuck::from_bytes(bytes)
    }
}
impl Record {
    pub fn new(barcode: u64, umi: u64, index: u64) -> Self {
        Self { barcode, umi, index }
    }
    pub fn as_bytes(&self) -> &[u8] {
        bytemuck::bytes_of(self)
    }
    pub fn from_bytes(bytes: &[u8]) -> Self {
        *bytemuck::from_bytes(bytes)
    }
}

// ---- Writer<W> (writer.rs) -------------------------------------------------------------------
unsafe extern "C" fn write_tramp<W: Write>(user: *mut c_void, data: *const u8, len: usize) -> i32 {
    let w = &mut *(user as *mut W);
    match w.write_all(std::slice::from_raw_parts(data, len)) {
        Ok(()) => 0,
        Err(e) => e.raw_os_error().unwrap_or(5),
    }
}
unsafe extern "C" fn flush_tramp<W: Write>(user: *mut c_void) -> i32 {
    match (&mut *(user as *mut W)).flush() {
        Ok(()) => 0,
        Err(e) => e.raw_os_error().unwrap_or(5),
    }
}

/// `Writer<W: Write>`: the sink is boxed so its address is stable for the C callbacks.
pub struct Writer<W: Write> {
    raw: *mut ffi::ibu_writer_t,
    inner: Box<W>,
}
impl<W: Write> Writer<W> {
    fn open(inner: W, header: Option<&Header>) -> Result<Self> {
        let mut inner = Box::new(inner);
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::ibu_writer_open_callback(write_tramp::<W>, Some(flush_tramp::<W>), (&mut *inner) as *mut W as *mut c_void,
                                          header.map_or(std::ptr::null(), |h| h as *const _), &mut raw)
        })?;
        Ok(Self { raw, inner })
    }
    pub fn new(inner: W, header: Header) -> Result<Self> {
        Self::open(inner, Some(&header))
    }
    pub fn new_headless(inner: W) -> Self {
        Self::open(inner, None).expect("headless open performs no I/O")
    }
    pub fn records_written(&self) -> u64 {
        unsafe { ffi::ibu_writer_records_written(self.raw) }
    }
    pub fn write_record(&mut self, record: &Record) -> Result<()> {
        check(unsafe { ffi::ibu_writer_write_record(self.raw, record) })
    }
    pub fn write_batch(&mut self, records: &[Record]) -> Result<()> {
        check(unsafe { ffi::ibu_writer_write_batch(self.raw, records.as_ptr(), records.len()) })
    }
    pub fn write_iter<I: Iterator<Item = Record>>(&mut self, records: I) -> Result<()> {
        for r in records {
            self.write_record(&r)?;
        }
        Ok(())
    }
    /// Device-resident AoS records -> pinned ring -> this writer (same buffered/direct rule).
    pub fn write_batch_device(&mut self, ctx: &device::Context, d_records: *const c_void, n: usize) -> Result<()> {
        check(unsafe {
            ffi::ibu_writer_write_batch_device(self.raw, ctx.raw, std::ptr::null(), d_records, n, std::ptr::null_mut())
        })
    }
    pub fn finish(&mut self) -> Result<()> {
        check(unsafe { ffi::ibu_writer_finish(self.raw) })
    }
    pub fn into_inner(self) -> W {
        // reference semantics: no flush (writer.rs:507-511)
        let this = std::mem::ManuallyDrop::new(self);
        unsafe {
            ffi::ibu_writer_into_inner(this.raw, std::ptr::null_mut(), std::ptr::null_mut());
            *std::ptr::read(&this.inner)
        }
    }
}
impl Writer<Vec<u8>> {
    // `ingest(&mut self, other: &mut Writer<Vec<u8>>)` (writer.rs:477-482): flush `other`, append its
    // bytes, clear them.  With a callback sink the bytes live in `other.inner`.
}
impl<W: Write> Writer<W> {
    pub fn ingest(&mut self, other: &mut Writer<Vec<u8>>) -> Result<()> {
        other.finish()?;
        let bytes = std::mem::take(&mut *other.inner);
        let recs: &[u8] = &bytes;
        check(unsafe { ffi::ibu_writer_write_batch(self.raw, recs.as_ptr() as *const Record, recs.len() / RECORD_SIZE) })
    }
}
impl<W: Write> Drop for Writer<W> {
    fn drop(&mut self) {
        unsafe { ffi::ibu_writer_close(self.raw) } // finish().ok()
    }
}
pub type BoxedWriter = Box<dyn Write + Send>;
impl Writer<BoxedWriter> {
    pub fn from_path<P: AsRef<Path>>(path: P, header: Header) -> Result<Self> {
        Self::new(Box::new(std::fs::File::create(path)?), header)
    }
    pub fn from_stdout(header: Header) -> Result<Self> {
        Self::new(Box::new(std::io::stdout()), header)
    }
    pub fn from_optional_path<P: AsRef<Path>>(path: Option<P>, header: Header) -> Result<Self> {
        match path {
            Some(p) => Self::from_path(p, header),
            None => Self::from_stdout(header),
        }
    }
}

// ---- Reader<R> (reader.rs) -----------------------------------------------------------------------
unsafe extern "C" fn read_tramp<R: Read>(user: *mut c_void, dst: *mut u8, cap: usize, got: *mut usize) -> i32 {
    match (&mut *(user as *mut R)).read(std::slice::from_raw_parts_mut(dst, cap)) {
        Ok(n) => {
            *got = n;
            0
        }
        Err(e) => e.raw_os_error().unwrap_or(5),
    }
}
pub struct Reader<R: Read> {
    raw: *mut ffi::ibu_reader_t,
    _inner: Box<R>,
}
impl<R: Read> Reader<R> {
    pub fn new(inner: R) -> Result<Self> {
        let mut inner = Box::new(inner);
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::ibu_reader_open_callback(read_tramp::<R>, (&mut *inner) as *mut R as *mut c_void, &mut raw) })?;
        Ok(Self { raw, _inner: inner })
    }
    pub fn read_batch(&mut self) -> Result<bool> {
        let mut has = 0;
        check(unsafe { ffi::ibu_reader_read_batch(self.raw, &mut has) })?;
        Ok(has != 0)
    }
    pub fn header(&self) -> Header {
        let mut h = bytemuck::Zeroable::zeroed();
        unsafe { ffi::ibu_reader_header(self.raw, &mut h) };
        h
    }
}
impl<R: Read> Iterator for Reader<R> {
    type Item = Result<Record>;
    fn next(&mut self) -> Option<Self::Item> {
        let (mut rec, mut got) = (Record::default(), 0);
        match check(unsafe { ffi::ibu_reader_next(self.raw, &mut rec, &mut got) }) {
            Err(e) => Some(Err(e)),
            Ok(()) if got != 0 => Some(Ok(rec)),
            Ok(()) => None,
        }
    }
}
impl<R: Read> Drop for Reader<R> {
    fn drop(&mut self) {
        unsafe { ffi::ibu_reader_close(self.raw) }
    }
}
/// `Reader::from_path` / `from_stdin` (reader.rs:345-434): the library sniffs gzip itself.
pub struct PathReader(*mut ffi::ibu_reader_t);
impl PathReader {
    pub fn from_path<P: AsRef<Path>>(path: P) -> Result<Self> {
        let c = CString::new(path.as_ref().to_string_lossy().as_bytes()).unwrap();
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::ibu_reader_open_path(c.as_ptr(), &mut raw) })?;
        Ok(Self(raw))
    }
    pub fn from_stdin() -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::ibu_reader_open_fd(0, &mut raw) })?;
        Ok(Self(raw))
    }
}
impl PathReader {
    /// Pull-style device stream over the rest of this reader (`ibu_stream_open_reader`): `Reader::read_batch` + `Iterator`
    /// (reader.rs:218-242, :279-306) with the batch in HBM.  The reader is borrowed mutably for as long as the stream lives.
    pub fn device_stream<'a>(&'a mut self, ctx: &'a device::Context) -> Result<device::DeviceStream<'a>> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::ibu_stream_open_reader(self.0, ctx.raw, std::ptr::null(), &mut raw) })?;
        Ok(device::DeviceStream { raw, _p: PhantomData })
    }
}
impl Iterator for PathReader {
    type Item = Result<Record>;
    fn next(&mut self) -> Option<Self::Item> {
        let (mut rec, mut got) = (Record::default(), 0);
        match check(unsafe { ffi::ibu_reader_next(self.0, &mut rec, &mut got) }) {
            Err(e) => Some(Err(e)),
            Ok(()) if got != 0 => Some(Ok(rec)),
            Ok(()) => None,
        }
    }
}
impl Drop for PathReader {
    fn drop(&mut self) {
        unsafe { ffi::ibu_reader_close(self.0) }
    }
}

pub fn load_to_vec<P: AsRef<Path>>(path: P) -> Result<(Header, Vec<Record>)> {
    let c = CString::new(path.as_ref().to_string_lossy().as_bytes()).unwrap();
    let (mut h, mut p, mut n) = (bytemuck::Zeroable::zeroed(), std::ptr::null_mut(), 0usize);
    check(unsafe { ffi::ibu_load_to_vec(c.as_ptr(), &mut h, &mut p, &mut n) })?;
    let v = unsafe { std::slice::from_raw_parts(p, n) }.to_vec();
    unsafe { ffi::ibu_free(p as *mut c_void) };
    Ok((h, v))
}

// ---- parallel (parallel.rs, mmap.rs) -----------------------------------------------------------------
pub trait ParallelProcessor: Send + Clone {
    fn process_record(&mut self, record: Record) -> Result<()>;
    fn on_batch_complete(&mut self) -> Result<()> {
        Ok(())
    }
    fn set_tid(&mut self, _tid: usize) {}
    fn get_tid(&self) -> Option<usize> {
        None
    }
}
pub trait ParallelReader {
    fn process_parallel<P: ParallelProcessor + Clone + 'static>(&self, processor: P, num_threads: usize) -> Result<()>;
}

pub struct MmapReader {
    raw: *mut ffi::ibu_mmap_t,
}
unsafe impl Send for MmapReader {}
impl Clone for MmapReader {
    fn clone(&self) -> Self {
        let mut out = std::ptr::null_mut();
        unsafe { ffi::ibu_mmap_clone(self.raw, &mut out) };
        Self { raw: out }
    }
}
impl MmapReader {
    pub fn new<P: AsRef<Path>>(path: P) -> Result<Self> {
        let c = CString::new(path.as_ref().to_string_lossy().as_bytes()).unwrap();
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::ibu_mmap_open(c.as_ptr(), &mut raw) })?;
        Ok(Self { raw })
    }
    #[allow(clippy::len_without_is_empty)]
    pub fn len(&self) -> usize {
        unsafe { ffi::ibu_mmap_len(self.raw) }
    }
    pub fn header(&self) -> Header {
        let mut h = bytemuck::Zeroable::zeroed();
        unsafe { ffi::ibu_mmap_header(self.raw, &mut h) };
        h
    }
    pub fn slice(&self, start: usize, end: usize) -> Result<&[Record]> {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { ffi::ibu_mmap_slice(self.raw, start, end, &mut p, &mut n) })?;
        Ok(unsafe { std::slice::from_raw_parts(p, n) })
    }
    /// One shard of the static split on one GPU (the device form of process_parallel).
    pub fn process_device_reduce(&self, ctx: &device::Context, shard: usize, n_shards: usize)
                                 -> Result<ffi::ibu_reduce_result_t> {
        let mut out = ffi::ibu_reduce_result_t::default();
        check(unsafe {
            ffi::ibu_mmap_process_device(self.raw, ctx.raw, std::ptr::null(), 1, shard, n_shards,
                                         &mut out as *mut _ as *mut c_void, std::ptr::null_mut())
        })?;
        Ok(out)
    }
    /// `process_parallel(processor, n)` with a GPU per worker, in ONE call (src/io/mmap.rs:286-332): worker i is a host
    /// thread + a context on `devices[i]` taking shard i of the static split; an empty list means every visible device
    /// (as `num_threads == 0` means every core).  Returns the total (count, wrapping sums, XORs) and the per-device
    /// partials; the first error in worker order is the call's error (quirk Q12).
    pub fn process_devices(&self, devices: &[i32]) -> Result<(ffi::ibu_reduce_result_t, Vec<ffi::ibu_reduce_result_t>)> {
        let n = if devices.is_empty() { device::device_count()? as usize } else { devices.len() };
        let mut total = ffi::ibu_reduce_result_t::default();
        let mut parts = vec![ffi::ibu_reduce_result_t::default(); n.max(1)];
        check(unsafe {
            ffi::ibu_mmap_process_devices(self.raw, if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() },
                                          devices.len(), std::ptr::null(), 1, parts.as_mut_ptr() as *mut c_void,
                                          &mut total, std::ptr::null_mut())
        })?;
        parts.truncate(n);
        Ok((total, parts))
    }
}
impl MmapReader {
    /// Pull-style device stream over one shard of the static split (`ibu_stream_open_mmap`): the per-batch loop of
    /// `process_parallel` (mmap.rs:312-320) with every batch resident in HBM.
    pub fn device_stream<'a>(&'a self, ctx: &'a device::Context, shard: usize, n_shards: usize) -> Result<device::DeviceStream<'a>> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::ibu_stream_open_mmap(self.raw, ctx.raw, std::ptr::null(), shard, n_shards, &mut raw) })?;
        Ok(device::DeviceStream { raw, _p: PhantomData })
    }
}
impl Drop for MmapReader {
    fn drop(&mut self) {
        unsafe { ffi::ibu_mmap_close(self.raw) }
    }
}

struct ProcBox<P: ParallelProcessor> {
    p: P,
    err: Option<IbuError>,
}
unsafe extern "C" fn p_clone<P: ParallelProcessor>(u: *mut c_void) -> *mut c_void {
    let src = &*(u as *const ProcBox<P>);
    Box::into_raw(Box::new(ProcBox { p: src.p.clone(), err: None })) as *mut c_void
}
unsafe extern "C" fn p_drop<P: ParallelProcessor>(u: *mut c_void) {
    drop(Box::from_raw(u as *mut ProcBox<P>))
}
unsafe extern "C" fn p_rec<P: ParallelProcessor>(u: *mut c_void, r: *const Record) -> i32 {
    let b = &mut *(u as *mut ProcBox<P>);
    match b.p.process_record(*r) {
        Ok(()) => 0,
        Err(e) => {
            b.err = Some(e);
            1
        }
    }
}
unsafe extern "C" fn p_batch<P: ParallelProcessor>(u: *mut c_void) -> i32 {
    let b = &mut *(u as *mut ProcBox<P>);
    match b.p.on_batch_complete() {
        Ok(()) => 0,
        Err(e) => {
            b.err = Some(e);
            1
        }
    }
}
impl ParallelReader for MmapReader {
    fn process_parallel<P: ParallelProcessor + Clone + 'static>(&self, processor: P, num_threads: usize) -> Result<()> {
        let vt = ffi::ibu_processor_vtable_t {
            clone: Some(p_clone::<P>),
            drop: Some(p_drop::<P>),
            process_record: Some(p_rec::<P>),
            on_batch_complete: Some(p_batch::<P>),
            set_tid: None, // never called by the reference either (mmap.rs:308-322)
        };
        let mut root = ProcBox { p: processor, err: None };
        check(unsafe { ffi::ibu_mmap_process_parallel(self.raw, &vt, &mut root as *mut _ as *mut c_void, num_threads) })
    }
}

// ---- device side ------------------------------------------------------------------------------------------
pub mod device {
    use super::*;
    /// Visible HIP devices (`ibu_device_count`).
    pub fn device_count() -> Result<i32> {
        let mut n = 0i32;
        check(unsafe { ffi::ibu_device_count(&mut n) })?;
        Ok(n)
    }
    pub struct Context {
        pub(crate) raw: *mut ffi::ibu_ctx_t,
    }
    pub struct DeviceBuf<'c> {
        pub ptr: *mut c_void,
        pub bytes: usize,
        ctx: &'c Context,
        _p: PhantomData<&'c ()>,
    }
    impl Context {
        pub fn new(device: i32) -> Result<Self> {
            let mut raw = std::ptr::null_mut();
            check(unsafe { ffi::ibu_ctx_create(device, &mut raw) })?;
            Ok(Self { raw })
        }
        /// `ibu_ctx_set_option`: `"base_order"` (bit order of the 2-bit codec), `"alloc_probe_tries"` (placement probing for the
        /// arrays of 256 MiB and more that the library allocates: `alloc`, the destination of `load_to_device`), `"blocks_per_cu"`,
        /// the sort's A/B switches; unknown keys and out-of-range values are `InvalidArg`.
        pub fn set_option(&self, key: &str, value: i64) -> Result<()> {
            let k = CString::new(key).unwrap();
            check(unsafe { ffi::ibu_ctx_set_option(self.raw, k.as_ptr(), value) })
        }
        /// Device memory from the library (placement-probed under option `"alloc_probe_tries"` from 256 MiB on).
        pub fn alloc(&self, bytes: usize) -> Result<DeviceBuf<'_>> {
            let mut p = std::ptr::null_mut();
            check(unsafe { ffi::ibu_device_alloc(self.raw, bytes, &mut p) })?;
            Ok(DeviceBuf { ptr: p, bytes, ctx: self, _p: PhantomData })
        }
        /// For arrays that stay resident: up to `tries` candidate allocations, the one a write + read streams over
        /// fastest is kept (placement probing, `ibu_device_alloc_probed`).
        pub fn alloc_probed(&self, bytes: usize, tries: u32) -> Result<(DeviceBuf<'_>, ffi::ibu_alloc_probe_t)> {
            let mut p = std::ptr::null_mut();
            let mut rep = ffi::ibu_alloc_probe_t::default();
            check(unsafe { ffi::ibu_device_alloc_probed(self.raw, bytes, tries, &mut p, &mut rep) })?;
            Ok((DeviceBuf { ptr: p, bytes, ctx: self, _p: PhantomData }, rep))
        }
        /// K2: AoS records -> barcode ASCII, UMI ASCII, index column (async on the context stream).
        pub fn decode_ascii(&self, recs: &DeviceBuf, n: usize, h: &Header, bc: &DeviceBuf, umi: &DeviceBuf, idx: &DeviceBuf) -> Result<()> {
            check(unsafe {
                ffi::ibu_decode_ascii(self.raw, recs.ptr, n, h.bc_len, h.umi_len, bc.ptr as *mut u8, umi.ptr as *mut u8,
                                      idx.ptr as *mut u64, std::ptr::null_mut())
            })
        }
        /// K3 + status: columns -> AoS records; Err(InvalidBase) if any byte is outside ACGTacgt.
        pub fn encode_ascii(&self, bc: &DeviceBuf, umi: &DeviceBuf, idx: &DeviceBuf, n: usize, h: &Header, recs: &DeviceBuf) -> Result<()> {
            check(unsafe {
                ffi::ibu_encode_ascii(self.raw, bc.ptr as *const u8, umi.ptr as *const u8, idx.ptr as *const u64, 0, n,
                                      h.bc_len, h.umi_len, recs.ptr, std::ptr::null_mut())
            })?;
            check(unsafe { ffi::ibu_codec_status(self.raw, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut()) })
        }
        pub fn reduce(&self, recs: &DeviceBuf, n: usize) -> Result<ffi::ibu_reduce_result_t> {
            let mut out = ffi::ibu_reduce_result_t::default();
            check(unsafe { ffi::ibu_reduce_reset(self.raw, std::ptr::null_mut()) })?;
            check(unsafe { ffi::ibu_reduce(self.raw, recs.ptr, n, std::ptr::null_mut()) })?;
            check(unsafe { ffi::ibu_reduce_fetch(self.raw, std::ptr::null_mut(), &mut out) })?;
            Ok(out)
        }
        /// Streaming device-to-device copy (the memcpy of `Writer::write_slice` / `ingest` for HBM-resident batches).
        pub fn copy(&self, dst: &DeviceBuf, src: &DeviceBuf, bytes: usize) -> Result<()> {
            check(unsafe { ffi::ibu_device_copy(self.raw, dst.ptr, src.ptr, bytes, std::ptr::null_mut()) })
        }
        /// Sort by (barcode, umi, index) — `Record`'s derived `Ord`; `tmp` is n*24 bytes of scratch.
        pub fn sort_records(&self, recs: &DeviceBuf, tmp: &DeviceBuf, n: usize) -> Result<()> {
            check(unsafe { ffi::ibu_sort_records(self.raw, recs.ptr, tmp.ptr, n, std::ptr::null_mut()) })
        }
        /// The same order over several shards, one per context (= per GPU), in one call (`ibu_sort_records_contexts`): shard i
        /// ends up with the i-th range of the global order; returns the new record counts.  `shards[i]` = (records, tmp, n,
        /// capacity in records) on `ctxs[i]`'s device.  Leave headroom in `capacity` (up to 32 shards the owners are cut at the
        /// boundaries of 256 sampled key ranges: shares differ by up to total / 256 — 3 % of a share with 8 shards, 12 % with 32;
        /// beyond that, and when such a cut does not fit, at sampled quantiles: a few percent); a shard that would overflow fails the call with every shard still holding its own records.  Experimental: never
        /// run on two distinct GPUs.
        pub fn sort_records_contexts(ctxs: &[&Context], shards: &[(&DeviceBuf, &DeviceBuf, usize, usize)]) -> Result<Vec<usize>> {
            assert_eq!(ctxs.len(), shards.len());
            let raw: Vec<*mut ffi::ibu_ctx_t> = ctxs.iter().map(|c| c.raw).collect();
            let mut sh: Vec<ffi::ibu_sort_shard_t> = shards
                .iter()
                .map(|(r, t, n, cap)| ffi::ibu_sort_shard_t { d_records: r.ptr, d_tmp: t.ptr, n: *n, capacity: *cap })
                .collect();
            check(unsafe { ffi::ibu_sort_records_contexts(raw.as_ptr(), raw.len(), sh.as_mut_ptr()) })?;
            Ok(sh.iter().map(|s| s.n).collect())
        }
        /// `a[..n] == b[..n]` on device-resident records (`Record: PartialEq`), with the position: the index of the
        /// first record that differs, `n` if none does.
        pub fn first_mismatch(&self, a: &DeviceBuf, b: &DeviceBuf, n: usize) -> Result<usize> {
            let mut f = 0u64;
            check(unsafe { ffi::ibu_records_first_mismatch(self.raw, a.ptr, b.ptr, n, &mut f, std::ptr::null_mut()) })?;
            Ok(f as usize)
        }
        /// OR / AND census of n device records: `[or; 3], [and; 3], index_drops, order_drops` (ibu_records_census).
        pub fn census(&self, recs: &DeviceBuf, n: usize) -> Result<[u64; 8]> {
            let mut c = [0u64; 8];
            check(unsafe { ffi::ibu_records_census(self.raw, recs.ptr, n, c.as_mut_ptr(), std::ptr::null_mut()) })?;
            Ok(c)
        }
        /// Plan of the compacted keys for records with these OR / AND words (ibu_key_plan_init); `plan.k` = varying bytes.
        pub fn key_plan(or_words: &[u64; 3], and_words: &[u64; 3]) -> Result<ffi::ibu_key_plan_t> {
            let mut p = std::mem::MaybeUninit::<ffi::ibu_key_plan_t>::uninit();
            check(unsafe { ffi::ibu_key_plan_init(or_words.as_ptr(), and_words.as_ptr(), p.as_mut_ptr()) })?;
            Ok(unsafe { p.assume_init() })
        }
        /// n records -> n 12-byte elements (plan.k <= 12) and back: the exchange format of the multi-GPU sort.
        pub fn compact(&self, plan: &ffi::ibu_key_plan_t, recs: &DeviceBuf, n: usize, elems: &DeviceBuf) -> Result<()> {
            check(unsafe { ffi::ibu_records_compact(self.raw, plan, recs.ptr, n, elems.ptr, std::ptr::null_mut()) })
        }
        pub fn expand(&self, plan: &ffi::ibu_key_plan_t, elems: &DeviceBuf, n: usize, recs: &DeviceBuf) -> Result<()> {
            check(unsafe { ffi::ibu_records_expand(self.raw, plan, elems.ptr, n, recs.ptr, std::ptr::null_mut()) })
        }
        pub fn is_sorted(&self, recs: &DeviceBuf, n: usize) -> Result<bool> {
            let mut s = 0i32;
            check(unsafe { ffi::ibu_is_sorted(self.raw, recs.ptr, n, std::ptr::null_mut(), &mut s) })?;
            Ok(s != 0)
        }
        /// The doc example's `BarcodeAnalyzer` (parallel.rs:72-98) on sorted device records:
        /// (barcode, records, distinct UMIs) per barcode in ascending barcode order.
        pub fn barcode_counts(&self, sorted: &DeviceBuf, n: usize) -> Result<Vec<(u64, u64, u64)>> {
            let (mut nb, mut np) = (0usize, 0usize);
            check(unsafe {
                ffi::ibu_barcode_counts(self.raw, sorted.ptr, n, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(),
                                        0, &mut nb, &mut np, std::ptr::null_mut())
            })?;
            if nb == 0 {
                return Ok(Vec::new());
            }
            let (b, c, u) = (self.alloc(8 * nb)?, self.alloc(8 * nb)?, self.alloc(8 * nb)?);
            check(unsafe {
                ffi::ibu_barcode_counts(self.raw, sorted.ptr, n, b.ptr as *mut u64, c.ptr as *mut u64, u.ptr as *mut u64, nb,
                                        &mut nb, &mut np, std::ptr::null_mut())
            })?;
            let mut host = vec![0u64; 3 * nb];
            for (k, d) in [&b, &c, &u].iter().enumerate() {
                check(unsafe {
                    ffi::ibu_memcpy_d2h(self.raw, host[k * nb..].as_mut_ptr() as *mut c_void, d.ptr, 8 * nb, std::ptr::null_mut())
                })?;
            }
            check(unsafe { ffi::ibu_ctx_synchronize(self.raw, std::ptr::null_mut()) })?;
            Ok((0..nb).map(|k| (host[k], host[nb + k], host[2 * nb + k])).collect())
        }
        /// Device analogue of `load_to_vec`.
        pub fn load_to_device<P: AsRef<Path>>(&self, path: P) -> Result<(Header, *mut c_void, usize)> {
            let c = CString::new(path.as_ref().to_string_lossy().as_bytes()).unwrap();
            let (mut h, mut p, mut n) = (bytemuck::Zeroable::zeroed(), std::ptr::null_mut(), 0usize);
            check(unsafe { ffi::ibu_load_to_device(self.raw, c.as_ptr(), std::ptr::null(), &mut h, &mut p, 0, &mut n, std::ptr::null_mut()) })?;
            Ok((h, p, n))
        }
        /// The same for a BGZF (bgzip) file of the records, inflated on the device: the compressed bytes cross the link.
        pub fn load_bgzf_to_device<P: AsRef<Path>>(&self, path: P) -> Result<(Header, *mut c_void, usize)> {
            let c = CString::new(path.as_ref().to_string_lossy().as_bytes()).unwrap();
            let (mut h, mut p, mut n) = (bytemuck::Zeroable::zeroed(), std::ptr::null_mut(), 0usize);
            check(unsafe { ffi::ibu_load_bgzf_to_device(self.raw, c.as_ptr(), std::ptr::null(), &mut h, &mut p, 0, &mut n, std::ptr::null_mut()) })?;
            Ok((h, p, n))
        }
        /// Shard `shard` of `n_shards` of a BGZF file's records (the split of `process_parallel`): `(header, device pointer, n, first record)`.
        pub fn load_bgzf_shard_to_device<P: AsRef<Path>>(&self, path: P, shard: usize, n_shards: usize) -> Result<(Header, *mut c_void, usize, u64)> {
            let c = CString::new(path.as_ref().to_string_lossy().as_bytes()).unwrap();
            let (mut h, mut p, mut n, mut first) = (bytemuck::Zeroable::zeroed(), std::ptr::null_mut(), 0usize, 0u64);
            check(unsafe {
                ffi::ibu_load_bgzf_shard_to_device(self.raw, c.as_ptr(), std::ptr::null(), shard, n_shards, &mut h, &mut p, 0, &mut n, &mut first, std::ptr::null_mut())
            })?;
            Ok((h, p, n, first))
        }
    }
    /// `ibu_stream_t`: iterate to pull one device-resident batch at a time.  An `Err` item is the source's error
    /// (`TruncatedRecord`, `Io`, `Niffler` ...), delivered after the batches in front of it; iteration ends after it.
    pub struct DeviceStream<'a> {
        pub(crate) raw: *mut ffi::ibu_stream_t,
        pub(crate) _p: PhantomData<&'a ()>,
    }
    /// One batch: `n` records of 24 bytes at `d_records` (a device ring slot), the first of them record number `first_index`.
    /// Dropping it gives the slot back: it is refilled once the work queued so far on the context's stream has run.
    pub struct DeviceBatch<'s> {
        pub d_records: *const c_void,
        pub n: usize,
        pub first_index: u64,
        stream: *mut ffi::ibu_stream_t,
        _p: PhantomData<&'s ()>,
    }
    impl<'a> DeviceStream<'a> {
        pub fn header(&self) -> Result<Header> {
            let mut h: Header = bytemuck::Zeroable::zeroed();
            check(unsafe { ffi::ibu_stream_header(self.raw, &mut h) })?;
            Ok(h)
        }
        pub fn stats(&self) -> Result<ffi::ibu_stream_stats_t> {
            let mut st = ffi::ibu_stream_stats_t::default();
            check(unsafe { ffi::ibu_stream_stats(self.raw, &mut st) })?;
            Ok(st)
        }
        /// The next batch, to be read on `hip_stream` (null: the context's stream); `Ok(None)` at the end.
        pub fn next_on(&mut self, hip_stream: *mut c_void) -> Result<Option<DeviceBatch<'_>>> {
            let (mut d, mut n, mut first) = (std::ptr::null(), 0usize, 0u64);
            check(unsafe { ffi::ibu_stream_next(self.raw, hip_stream, &mut d, &mut n, &mut first) })?;
            Ok(if n == 0 { None } else { Some(DeviceBatch { d_records: d, n, first_index: first, stream: self.raw, _p: PhantomData }) })
        }
    }
    impl<'a> Iterator for DeviceStream<'a> {
        type Item = Result<DeviceBatch<'a>>;
        fn next(&mut self) -> Option<Self::Item> {
            let (mut d, mut n, mut first) = (std::ptr::null(), 0usize, 0u64);
            match check(unsafe { ffi::ibu_stream_next(self.raw, std::ptr::null_mut(), &mut d, &mut n, &mut first) }) {
                Err(e) => Some(Err(e)),
                Ok(()) if n != 0 => Some(Ok(DeviceBatch { d_records: d, n, first_index: first, stream: self.raw, _p: PhantomData })),
                Ok(()) => None,
            }
        }
    }
    impl Drop for DeviceBatch<'_> {
        fn drop(&mut self) {
            unsafe { ffi::ibu_stream_release(self.stream, self.d_records, std::ptr::null_mut()) };
        }
    }
    impl Drop for DeviceStream<'_> {
        fn drop(&mut self) {
            unsafe { ffi::ibu_stream_close(self.raw) }
        }
    }
    impl Context {
        /// Where the device hangs off the host and where the pinned ring landed (`ibu_ctx_numa`, option `"numa"`).
        pub fn numa(&self) -> Result<ffi::ibu_numa_info_t> {
            let mut i: ffi::ibu_numa_info_t = unsafe { std::mem::zeroed() };
            check(unsafe { ffi::ibu_ctx_numa(self.raw, &mut i) })?;
            Ok(i)
        }
    }
    impl Drop for Context {
        fn drop(&mut self) {
            unsafe { ffi::ibu_ctx_destroy(self.raw) }
        }
    }
    impl Drop for DeviceBuf<'_> {
        fn drop(&mut self) {
            unsafe { ffi::ibu_device_free(self.ctx.raw, self.ptr) };
        }
    }
}
