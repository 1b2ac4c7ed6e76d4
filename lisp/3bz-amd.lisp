;;;; 3bz-amd.lisp — CFFI shim: 3bz's exported API (package.lisp:13-27) over lib3bz_amd.so.
;;;;
;;;; Host code stays Common Lisp; this file is the thin layer BASELINE.json's north_star asks for.
;;;; It binds include/tbz_amd.h one-to-one and re-exports the same symbols 3bz exports, with the
;;;; same lambda lists and return values, so `(3bz-amd:decompress-vector v :format :zlib)` is a
;;;; drop-in for `(3bz:decompress-vector v :format :zlib)` on the octet-vector path.
;;;;
;;;; STATUS: written against the header; NOT loadable/testable in the build image (no Lisp
;;;; implementation exists there — SURVEY §8c).  The same surface, in Python over ctypes, is
;;;; 3bz_amd/api.py and is what the parity tests drive.  Keep the two in step.
;;;;
;;;; Resuming a state after input-underrun / output-overflow (3bz's chunked protocol,
;;;; deflate.lisp:114-137) is not implemented on the device path yet (SURVEY §8f-2): the second
;;;; DECOMPRESS call on an unfinished state signals an error instead of silently falling back.

(defpackage #:3bz-amd
  (:use #:cl)
  (:export #:decompress #:decompress-vector
           #:make-octet-vector-context
           #:make-deflate-state #:make-zlib-state #:make-gzip-state
           #:finished #:input-underrun #:output-overflow
           #:replace-output-buffer
           ;; engine management (no counterpart in 3bz: a deflate-state is self-contained)
           #:*engine* #:open-engine #:close-engine #:with-engine))
(in-package #:3bz-amd)

(cffi:define-foreign-library lib3bz-amd
  (:unix (:or "lib3bz_amd.so" "./3bz_amd/lib3bz_amd.so"))
  (t (:default "lib3bz_amd")))
(cffi:use-foreign-library lib3bz-amd)

;;; struct tbz_result (64 octets) — include/tbz_amd.h
(cffi:defcstruct tbz-result
  (status :int32) (segments :uint32)
  (out-len :uint64) (out-total :uint64) (in-consumed :uint64)
  (adler32 :uint32) (crc32 :uint32) (trailer-check :uint32) (trailer-isize :uint32)
  (flags :uint32) (reserved :uint32) (boundary-out :uint64))

(cffi:defcfun ("tbz_ctx_create" %ctx-create) :int (device :int) (out :pointer))
(cffi:defcfun ("tbz_ctx_destroy" %ctx-destroy) :void (ctx :pointer))
(cffi:defcfun ("tbz_strerror" %strerror) :string (code :int))
(cffi:defcfun ("tbz_last_error" %last-error) :string (ctx :pointer))
(cffi:defcfun ("tbz_inflate" %inflate) :int
  (ctx :pointer) (format :int) (in :pointer) (in-len :size) (out :pointer) (out-cap :size) (res :pointer))
(cffi:defcfun ("tbz_inflate_size" %inflate-size) :int
  (ctx :pointer) (format :int) (in :pointer) (in-len :size) (res :pointer))

(deftype octet () '(unsigned-byte 8))
(deftype octet-vector () '(simple-array octet (*)))

(defvar *engine* nil "the tbz_ctx used by DECOMPRESS / DECOMPRESS-VECTOR")

(defun open-engine (&optional (device 0))
  (cffi:with-foreign-object (p :pointer)
    (let ((r (%ctx-create device p)))
      (unless (zerop r) (error "tbz_ctx_create: ~a" (%strerror r)))
      (setf *engine* (cffi:mem-ref p :pointer)))))
(defun close-engine ()
  (when *engine* (%ctx-destroy *engine*) (setf *engine* nil)))
(defmacro with-engine ((&optional (device 0)) &body body)
  `(let ((*engine* nil)) (open-engine ,device) (unwind-protect (progn ,@body) (close-engine))))
(defun engine () (or *engine* (open-engine)))

(defun format-code (format)
  (ecase format (:deflate 0) (:zlib 1) (:gzip 2)))  ; api.lisp:31-34

;;; io-common.lisp:36-45 — octet-vector-context + context-boxes
(defstruct (octet-vector-context (:constructor %make-ovc))
  octet-vector (start 0) (end 0) (offset 0))
(defun make-octet-vector-context (vector &key (start 0) (offset start) (end (length vector)))
  (%make-ovc :octet-vector vector :start start :end end :offset offset))

;;; deflate.lisp:4-62 / zlib.lisp:3-12 / gzip.lisp:3-28 — the observable slots
(defstruct (deflate-state (:conc-name ds-))
  (output-buffer (make-array 0 :element-type 'octet) :type octet-vector)
  (output-offset 0 :type fixnum)
  (finished nil) (output-overflow nil) (input-underrun nil)
  (calls 0 :type fixnum))
(defstruct (zlib-state (:include deflate-state)))
(defstruct (gzip-state (:include deflate-state)))

(defun finished (state) (ds-finished state))                  ; api.lisp:67-68
(defun input-underrun (state) (ds-input-underrun state))      ; api.lisp:69-70
(defun output-overflow (state) (ds-output-overflow state))    ; api.lisp:71-72

(defun replace-output-buffer (state buffer)                   ; api.lisp:12-21
  (unless (or (zerop (ds-output-offset state)) (ds-output-overflow state))
    (error "can't switch buffers without filling old one yet."))
  (setf (ds-output-buffer state) buffer
        (ds-output-offset state) 0
        (ds-output-overflow state) nil))

(defun state-format (state)
  (etypecase state (gzip-state 2) (zlib-state 1) (deflate-state 0)))

(defun %call-inflate (format vector start end out)
  "pin both vectors (the pattern of 3bz's own bench.lisp:61) and run one tbz_inflate"
  (cffi:with-foreign-object (res '(:struct tbz-result))
    (cffi:with-pointer-to-vector-data (pin vector)
      (cffi:with-pointer-to-vector-data (pout out)
        (let ((r (%inflate (engine) format (cffi:inc-pointer pin start) (- end start)
                           pout (length out) res)))
          (unless (zerop r) (error "tbz_inflate: ~a: ~a" (%strerror r) (%last-error (engine)))))))
    (cffi:with-foreign-slots ((status out-len out-total in-consumed flags) res (:struct tbz-result))
      (values status out-len out-total in-consumed flags))))

(defun decompress (context state)                              ; api.lisp:3-10
  (when (and (plusp (ds-calls state)) (not (ds-finished state)))
    (error "resuming a stream (chunked input/output) is not implemented on the device path"))
  (incf (ds-calls state))
  (setf (ds-input-underrun state) nil (ds-output-overflow state) nil)
  (multiple-value-bind (status out-len out-total in-consumed flags)
      (%call-inflate (state-format state)
                     (octet-vector-context-octet-vector context)
                     (octet-vector-context-offset context)
                     (octet-vector-context-end context)
                     (ds-output-buffer state))
    (declare (ignore out-total))
    (when (minusp status) (error "~a" (%strerror status)))     ; Lisp conditions of the reference
    (setf (ds-finished state) (= status 0)
          (ds-input-underrun state) (= status 1)
          (ds-output-overflow state) (= status 2)
          (ds-output-offset state) out-len)
    (if (ds-finished state)
        (incf (octet-vector-context-offset context) in-consumed)
        (setf (octet-vector-context-offset context) (octet-vector-context-end context)))
    ;; gzip: final block decoded but crc32/ISIZE cut off => (return-from decompress-gzip 0)
    (if (and (= status 1) (typep state 'gzip-state) (logbitp 1 flags))
        0
        out-len)))

(defun decompress-vector (compressed &key (format :zlib) (start 0) (end (length compressed)) output)
  "api.lisp:23-65.  Returns (values buffer count)."
  (let ((fmt (format-code format)))
    (flet ((check (status)
             (when (minusp status) (error "~a" (%strerror status)))
             (unless (= status 0)
               (if (= status 1)
                   (error "incomplete ~a stream" format)                      ; api.lisp:43-44
                   (error "not enough space to decompress ~a stream" format)))))  ; api.lisp:45-46
      (if output
          (multiple-value-bind (status out-len) (%call-inflate fmt compressed start end output)
            (check status)
            (values output out-len))
          ;; the reference grows 32 KiB buffers by doubling and gathers (api.lisp:48-65);
          ;; the engine's count pass gives the size, so allocate exactly once
          (cffi:with-foreign-object (res '(:struct tbz-result))
            (cffi:with-pointer-to-vector-data (pin compressed)
              (let ((r (%inflate-size (engine) fmt (cffi:inc-pointer pin start) (- end start) res)))
                (unless (zerop r) (error "tbz_inflate_size: ~a" (%strerror r)))))
            (let ((status (cffi:foreign-slot-value res '(:struct tbz-result) 'status))
                  (total (cffi:foreign-slot-value res '(:struct tbz-result) 'out-total)))
              (when (minusp status) (error "~a" (%strerror status)))
              (when (= status 1) (error "incomplete ~a stream" format))       ; api.lisp:55 assert
              (let ((buf (make-array total :element-type 'octet)))
                (multiple-value-bind (status2 out-len) (%call-inflate fmt compressed start end buf)
                  (check status2)
                  (values buf out-len)))))))))
