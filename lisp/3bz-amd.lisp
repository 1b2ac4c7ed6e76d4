;;;; 3bz-amd.lisp — CFFI shim: 3bz's exported API (package.lisp:13-27) over lib3bz_amd.so.
;;;;
;;;; Host code stays Common Lisp; this file is the thin layer BASELINE.json's north_star asks for.
;;;; It binds include/tbz_amd.h one-to-one and re-exports the same 14 symbols 3bz exports, with the
;;;; same lambda lists and return values, so `(3bz-amd:decompress-vector v :format :zlib)` is a
;;;; drop-in for `(3bz:decompress-vector v :format :zlib)`.
;;;;
;;;; STATUS: written against the header; NOT loadable in the build image (no Lisp implementation
;;;; exists there — SURVEY §8c).  tests/test_lisp_shim.py parses the defcstruct / defcfun / export
;;;; forms below and checks them against include/tbz_amd.h, the ctypes binding and the reference's
;;;; package.lisp; the same surface in Python over ctypes is 3bz_amd/api.py, which the parity
;;;; tests drive.  Keep the two in step.
;;;;
;;;; The chunked protocol (more input after input-underrun, a new buffer after output-overflow:
;;;; deflate.lisp:114-137, api.lisp:12-21) runs on a tbz_session: the resumable part of a
;;;; deflate-state lives on the device.

(defpackage #:3bz-amd
  (:use #:cl)
  (:export
   ;; ---- the reference's 14 (package.lisp:13-27)
   #:decompress
   #:decompress-vector
   #:with-octet-pointer
   #:make-octet-vector-context
   #:make-octet-stream-context
   #:make-octet-pointer-context
   #:make-deflate-state
   #:make-zlib-state
   #:make-gzip-state
   #:finished
   #:input-underrun
   #:output-overflow
   #:%resync-file-stream
   #:replace-output-buffer
   ;; ---- engine management (no counterpart in 3bz: a deflate-state is self-contained)
   #:*engine* #:open-engine #:close-engine #:with-engine #:trim-engine
   ;; ---- beyond the reference (SURVEY §8f-4): every member of a multi-member .gz
   #:decompress-gzip-members
   ;; ---- beyond the reference (SURVEY §8e): many independent streams over the GPUs of one node, from ONE Lisp process
   #:*engines* #:open-engines #:close-engines #:decompress-vectors))
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

;;; struct tbz_gzip_header — include/tbz_amd.h
(cffi:defcstruct tbz-gzip-header
  (status :int32) (header-len :uint32)
  (cm :uint32) (flg :uint32) (mtime :uint32) (xfl :uint32) (os :uint32)
  (extra-off :uint32) (extra-len :uint32) (name-off :uint32) (name-len :uint32)
  (comment-off :uint32) (comment-len :uint32) (hcrc-present :uint32) (hcrc :uint32) (stage :uint32))

(cffi:defcfun ("tbz_ctx_create" %ctx-create) :int (device :int) (out :pointer))
(cffi:defcfun ("tbz_ctx_destroy" %ctx-destroy) :void (ctx :pointer))
(cffi:defcfun ("tbz_ctx_trim" %ctx-trim) :int (ctx :pointer))
(cffi:defcfun ("tbz_strerror" %strerror) :string (code :int))
(cffi:defcfun ("tbz_last_error" %last-error) :string (ctx :pointer))
(cffi:defcfun ("tbz_inflate" %inflate) :int
  (ctx :pointer) (format :int) (in :pointer) (in-len :size) (out :pointer) (out-cap :size) (res :pointer))
(cffi:defcfun ("tbz_inflate_alloc" %inflate-alloc) :int
  (ctx :pointer) (format :int) (in :pointer) (in-len :size) (alloc :pointer) (user :pointer) (res :pointer))
(cffi:defcfun ("tbz_inflate_batch" %inflate-batch) :int
  (ctx :pointer) (format :int) (n :size) (ins :pointer) (in-lens :pointer) (outs :pointer) (out-caps :pointer)
  (results :pointer))
(cffi:defcfun ("tbz_session_create" %session-create) :int (ctx :pointer) (format :int) (out :pointer))
(cffi:defcfun ("tbz_session_destroy" %session-destroy) :void (session :pointer))
(cffi:defcfun ("tbz_session_feed" %session-feed) :int
  (session :pointer) (in :pointer) (in-len :size) (in-on-device :int))
(cffi:defcfun ("tbz_session_decompress" %session-decompress) :int
  (session :pointer) (out :pointer) (out-cap :size) (res :pointer))
(cffi:defcfun ("tbz_gzip_header_parse" %gzip-header-parse) :int (in :pointer) (in-len :size) (out :pointer))
(cffi:defcfun ("tbz_device_count" %device-count) :int)
(cffi:defcfun ("tbz_memcpy_d2h" %memcpy-d2h) :int (ctx :pointer) (h-dst :pointer) (d-src :pointer) (bytes :size))
(cffi:defcfun ("tbz_inflate_size" %inflate-size) :int
  (ctx :pointer) (format :int) (in :pointer) (in-len :size) (res :pointer))
(cffi:defcfun ("tbz_inflate_batch_multi" %inflate-batch-multi) :int
  (ctxs :pointer) (n-ctx :size) (format :int) (n :size) (ins :pointer) (in-lens :pointer) (outs :pointer)
  (out-caps :pointer) (results :pointer))
(cffi:defcfun ("tbz_inflate_gzip_members" %inflate-gzip-members) :int
  (ctx :pointer) (in :pointer) (in-len :size) (alloc :pointer) (user :pointer) (max-members :size) (results :pointer)
  (member-in-off :pointer) (n-members :pointer))
;;; one flush-delimited stream over several decoders (SURVEY §8e row 2): the cut points, the seam proof, and the three
;;; steps in one call for the contexts of OPEN-ENGINES
(cffi:defcfun ("tbz_inflate_sharded_plan" %sharded-plan) :int
  (in :pointer) (in-len :size) (n-parts :size) (cuts :pointer))
(cffi:defcfun ("tbz_inflate_sharded_verdict" %sharded-verdict) :int
  (format :int) (in :pointer) (in-len :size) (n-parts :size) (cuts :pointer) (recs :pointer) (part-check :pointer)
  (out-offs :pointer) (total :pointer) (check :pointer) (in-consumed :pointer) (why :pointer))
(cffi:defcfun ("tbz_inflate_sharded_multi" %inflate-sharded-multi) :int
  (ctxs :pointer) (n-ctx :size) (format :int) (in :pointer) (in-len :size) (out :pointer) (out-cap :size) (res :pointer)
  (sharded :pointer))

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
(defun trim-engine () (when *engine* (%ctx-trim *engine*)))

(defun format-code (format)
  (ecase format (:deflate 0) (:zlib 1) (:gzip 2)))  ; api.lisp:31-34

(defun check-call (r what)
  (unless (zerop r) (error "~a: ~a: ~a" what (%strerror r) (%last-error (engine)))))

;;; ---- contexts -------------------------------------------------------------------------------
;;; io-common.lisp:8-14,36-45 — context-boxes (start, end, offset) + octet-vector-context
(defstruct (context-boxes (:conc-name cb-)) (start 0) (end 0) (offset 0))

(defclass octet-vector-context ()
  ((octet-vector :reader octet-vector :initarg :octet-vector)
   (boxes :reader boxes :initarg :boxes)))
(defun make-octet-vector-context (vector &key (start 0) (offset start) (end (length vector)))
  (make-instance 'octet-vector-context
                 :octet-vector vector
                 :boxes (make-context-boxes :start start :offset offset :end end)))

;;; io-common.lisp:47-69 — a file stream; the reference reads it octet by octet ("very slow", README.md:13);
;;; here what the boxes span is read in one READ-SEQUENCE and handed to the session
(defclass octet-stream-context ()
  ((octet-stream :reader octet-stream :initarg :octet-stream)
   (boxes :reader boxes :initarg :boxes)))
(defun make-octet-stream-context (file-stream &key (start 0) (offset 0) (end (file-length file-stream)))
  (make-instance 'octet-stream-context
                 :octet-stream file-stream
                 :boxes (make-context-boxes :start start :offset offset :end end)))
(defgeneric %resync-file-stream (context))
(defmethod %resync-file-stream (context) (declare (ignore context)))
(defmethod %resync-file-stream ((context octet-stream-context))
  (file-position (octet-stream context) (cb-offset (boxes context))))
(defun valid-octet-stream (os)
  (and (typep os 'stream) (subtypep (stream-element-type os) 'octet) (open-stream-p os) (input-stream-p os)))

;;; io-mmap.lisp:21-54 — foreign memory valid inside a dynamic scope.  :DEVICE T says the memory is HBM (a raw
;;; device pointer): the session copies it device to device.
(defclass octet-pointer ()
  ((base :reader base :initarg :base)
   (size :reader size :initarg :size)
   (scope :reader scope :initarg :scope)
   (device :reader device-p :initarg :device :initform nil)))
(defmacro with-octet-pointer ((var pointer size &key device) &body body)
  (let ((scope (gensym "SCOPE")))
    `(let* ((,scope (cons t ',var)))
       (unwind-protect
            (let ((,var (make-instance 'octet-pointer :base ,pointer :size ,size :scope ,scope :device ,device)))
              ,@body)
         (setf (car ,scope) nil)))))
(defun valid-octet-pointer (op)
  (and (car (scope op)) (not (cffi:null-pointer-p (base op))) (plusp (size op))))
(defclass octet-pointer-context ()
  ((op :reader op :initarg :op)
   (pointer :reader %pointer :initarg :pointer)
   (boxes :reader boxes :initarg :boxes)))
(defun make-octet-pointer-context (octet-pointer &key (start 0) (offset 0) (end (size octet-pointer)))
  (make-instance 'octet-pointer-context
                 :op octet-pointer
                 :pointer (base octet-pointer)
                 :boxes (make-context-boxes :start start :offset offset :end end)))

;;; ---- states ---------------------------------------------------------------------------------
;;; deflate.lisp:4-62 / zlib.lisp:3-12 / gzip.lisp:3-28 — the observable slots; the resumable part is the session
(defstruct (deflate-state (:conc-name ds-))
  (output-buffer (make-array 0 :element-type 'octet) :type octet-vector)
  (output-offset 0 :type fixnum)
  (finished nil) (output-overflow nil) (input-underrun nil)
  (session nil)               ; tbz_session*
  (fed 0 :type fixnum))       ; input octets given to the session so far
(defstruct (zlib-state (:include deflate-state)))
(defstruct (gzip-state (:include deflate-state) (:conc-name gs-))
  (compression-method nil) (flags nil) (extra nil) (name nil) (comment nil)
  (operating-system nil) (mtime/unix nil) (mtime/universal nil) (compression-level :default)
  (header-octets (make-array 16 :element-type 'octet :adjustable t :fill-pointer 0)) (header-parsed nil))

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

(defun ensure-session (state)
  (or (ds-session state)
      (cffi:with-foreign-object (p :pointer)
        (check-call (%session-create (engine) (state-format state) p) "tbz_session_create")
        (let ((s (cffi:mem-ref p :pointer)))
          ;; the device memory goes with the state
          #+sbcl (sb-ext:finalize state (lambda () (%session-destroy s)) :dont-save t)
          (setf (ds-session state) s)))))

;;; gzip header metadata (gzip.lisp:144-241), decoded on the host from the octets the state has been given
(defun note-gzip-header (state octets start end)
  (unless (gs-header-parsed state)
    (loop for i from start below end
          while (< (fill-pointer (gs-header-octets state)) 70000)
          do (vector-push-extend (aref octets i) (gs-header-octets state)))
    (let* ((hb (coerce (gs-header-octets state) 'octet-vector))
           (n (length hb)))
      (cffi:with-foreign-object (h '(:struct tbz-gzip-header))
        (cffi:with-pointer-to-vector-data (p hb)
          (%gzip-header-parse p n h))
        (cffi:with-foreign-slots ((status stage flg mtime xfl os extra-off extra-len name-off name-len
                                          comment-off comment-len)
                                  h (:struct tbz-gzip-header))
          ;; the reference fills the slots as it reads (gzip.lisp:123-241): `stage` says how far the octets reach
          (flet ((text (off len)
                   ;; the reference's own form (gzip.lisp:214-217, :236-239): with :ERRORP NIL babel substitutes
                   ;; what is not utf-8 and returns a string, so the iso-8859-1 branch behind it never runs
                   (babel:octets-to-string (subseq hb off (+ off len)) :encoding :utf-8 :errorp nil)))
            (when (>= stage 2)
              (setf (gs-compression-method state) :deflate
                    (gs-flags state) (append (when (logbitp 4 flg) '(:comment)) (when (logbitp 3 flg) '(:name))
                                             (when (logbitp 2 flg) '(:extra)) (when (logbitp 1 flg) '(:header-crc))
                                             (when (logbitp 0 flg) '(:text)))))
            (when (and (>= stage 3) (not (zerop mtime)))
              (setf (gs-mtime/unix state) mtime
                    (gs-mtime/universal state) (+ mtime (encode-universal-time 0 0 0 1 1 1970 0))))
            (when (>= stage 4)
              (setf (gs-compression-level state) (or (case xfl (2 :maximum) (4 :fastest)) xfl)
                    (gs-operating-system state)
                    (if (<= 0 os 13)
                        (aref #(:fat :amiga :vms :unix :vm/cms :atari-tos :hpfs :macintosh :z-system :cp/m :tops-20
                                :ntfs :qdos :acorn-riscos)
                              os)
                        (list :unknown os))))
            (when (and (>= stage 5) (logbitp 2 flg)) (setf (gs-extra state) (subseq hb extra-off (+ extra-off extra-len))))
            (when (and (>= stage 6) (logbitp 3 flg)) (setf (gs-name state) (text name-off name-len)))
            (when (and (>= stage 7) (logbitp 4 flg)) (setf (gs-comment state) (text comment-off comment-len)))
            (unless (= status 1) (setf (gs-header-parsed state) t))))))))

;;; ---- decompress (api.lisp:3-10) --------------------------------------------------------------
(defun feed-context (context state session)
  "hand the octets of the context from offset to end to the session, return how many"
  (let* ((b (boxes context))
         (n (- (cb-end b) (cb-offset b))))
    (when (plusp n)
      (etypecase context
        (octet-vector-context
         (let ((v (octet-vector context)))
           (when (typep state 'gzip-state) (note-gzip-header state v (cb-offset b) (cb-end b)))
           (cffi:with-pointer-to-vector-data (p v)   ; pinned for the call (the pattern of bench.lisp:61)
             (check-call (%session-feed session (cffi:inc-pointer p (cb-offset b)) n 0) "tbz_session_feed"))))
        (octet-pointer-context
         (assert (valid-octet-pointer (op context)))                  ; io-mmap.lisp:66
         ;; the gzip-state's metadata slots are filled from the octets the state is GIVEN, whatever memory holds them:
         ;; the head of the range (a header is at most ~70 KB) is copied into a Lisp vector — read in place from host
         ;; memory, fetched from the device for a device pointer
         (when (and (typep state 'gzip-state) (not (gs-header-parsed state)))
           (let* ((m (min n 70000))
                  (v (make-array m :element-type 'octet))
                  (src (cffi:inc-pointer (%pointer context) (cb-offset b))))
             (if (device-p (op context))
                 (cffi:with-pointer-to-vector-data (pv v)
                   (check-call (%memcpy-d2h (engine) pv src m) "tbz_memcpy_d2h"))
                 (dotimes (i m) (setf (aref v i) (cffi:mem-aref src :uint8 i))))
             (note-gzip-header state v 0 m)))
         (check-call (%session-feed session (cffi:inc-pointer (%pointer context) (cb-offset b)) n
                                    (if (device-p (op context)) 1 0))
                     "tbz_session_feed"))
        (octet-stream-context
         (assert (valid-octet-stream (octet-stream context)))         ; io.lisp:71
         (let ((v (make-array n :element-type 'octet)))
           (file-position (octet-stream context) (cb-offset b))
           (setf n (read-sequence v (octet-stream context)))
           (when (typep state 'gzip-state) (note-gzip-header state v 0 n))
           (cffi:with-pointer-to-vector-data (p v)
             (check-call (%session-feed session p n 0) "tbz_session_feed")))))
      (setf (cb-offset b) (+ (cb-offset b) n)))
    (max n 0)))

(defun decompress (context state)
  (let* ((session (ensure-session state))
         (b (boxes context))
         (start-offset (cb-offset b))
         (fed-before (ds-fed state)))
    (setf (ds-input-underrun state) nil (ds-output-overflow state) nil)
    (when (ds-finished state) (return-from decompress (ds-output-offset state)))
    (incf (ds-fed state) (feed-context context state session))
    (let* ((out (ds-output-buffer state))
           (off (ds-output-offset state))
           (room (max 0 (- (length out) off))))
      (cffi:with-foreign-object (res '(:struct tbz-result))
        (cffi:with-pointer-to-vector-data (pout out)
          (check-call (%session-decompress session (cffi:inc-pointer pout off) room res) "tbz_session_decompress"))
        (cffi:with-foreign-slots ((status out-len in-consumed flags) res (:struct tbz-result))
          (setf (ds-output-offset state) (+ off out-len))
          (when (minusp status) (error "~a" (%strerror status)))     ; the Lisp conditions of the reference
          (setf (ds-finished state) (= status 0)
                (ds-input-underrun state) (= status 1)
                (ds-output-overflow state) (= status 2))
          (when (ds-finished state)
            ;; the context stands just behind the stream, its trailer included
            (setf (cb-offset b) (+ start-offset (max 0 (- in-consumed fed-before))))
            (when (typep context 'octet-stream-context) (%resync-file-stream context)))
          ;; gzip: final block decoded but crc32/ISIZE cut off => (return-from decompress-gzip 0)
          (if (and (= status 1) (typep state 'gzip-state) (logbitp 1 flags))
              0
              (ds-output-offset state)))))))

;;; ---- decompress-vector (api.lisp:23-65) -----------------------------------------------------
;;; tbz_alloc_fn: the engine knows the size now and copies the octets into what this returns.  A Lisp vector is pinned
;;; only INSIDE with-pointer-to-vector-data, and the engine writes after the callback has returned: so the callback
;;; hands out foreign memory, and the octets move into a Lisp vector once the call is over (one host copy; a
;;; static-vectors vector would save it where that library is at hand).
(defvar *alloc-results* nil "list of (pointer . n), newest first")
(cffi:defcallback alloc-octets :pointer ((user :pointer) (n :size))
  (declare (ignore user))
  (let ((p (if (zerop n) (cffi:null-pointer) (cffi:foreign-alloc :uint8 :count n))))
    (push (cons p n) *alloc-results*)
    p))
(defun take-alloc-result (cell)
  "the Lisp vector of one (pointer . n), the foreign memory freed"
  (destructuring-bind (p . n) cell
    (let ((v (make-array n :element-type 'octet)))
      (unless (zerop n)
        (cffi:with-pointer-to-vector-data (pv v)
          (cffi:foreign-funcall "memcpy" :pointer pv :pointer p :size n :pointer))
        (cffi:foreign-free p)
        (setf (car cell) (cffi:null-pointer)))   ; (so that a clean-up pass does not free it again)
      v)))

(defun decompress-vector (compressed &key (format :zlib) (start 0) (end (length compressed)) output)
  "Returns (values buffer count)."
  (let ((fmt (format-code format)))
    (flet ((check (status)
             (when (minusp status) (error "~a" (%strerror status)))
             (unless (= status 0)
               (if (= status 1)
                   (error "incomplete ~a stream" format)                          ; api.lisp:43-44
                   (error "not enough space to decompress ~a stream" format)))))  ; api.lisp:45-46
      (cffi:with-foreign-object (res '(:struct tbz-result))
        (cffi:with-pointer-to-vector-data (pin compressed)
          (if output
              (cffi:with-pointer-to-vector-data (pout output)
                (check-call (%inflate (engine) fmt (cffi:inc-pointer pin start) (- end start) pout (length output) res)
                            "tbz_inflate"))
              ;; the reference grows 32 KiB buffers by doubling and gathers (api.lisp:48-65); the engine decodes
              ;; once and asks for the buffer when it knows the size
              (let ((*alloc-results* nil))
                (unwind-protect
                     (check-call (%inflate-alloc (engine) fmt (cffi:inc-pointer pin start) (- end start)
                                                 (cffi:callback alloc-octets) (cffi:null-pointer) res)
                                 "tbz_inflate_alloc")
                  (setf output (if *alloc-results*
                                   (take-alloc-result (first *alloc-results*))
                                   (make-array 0 :element-type 'octet)))))))
        (check (cffi:foreign-slot-value res '(:struct tbz-result) 'status))
        (values output (cffi:foreign-slot-value res '(:struct tbz-result) 'out-len))))))

;;; ---- every member of a multi-member .gz (SURVEY §8f-4; 3bz stops after the first: gzip.lisp:277-286) ----
(defun decompress-gzip-members (compressed &key (start 0) (end (length compressed)) (max-members 1048576))
  "a list of octet vectors, one per member: each is what (decompress-vector v :format :gzip :start member-offset)
returns.  ONE call of tbz_inflate_gzip_members: the member starts are found on the device, all members are decoded as
one batch, and the walk that proves them runs inside the library.  A damaged or incomplete member signals what the
one-member call at its offset signals."
  ;; (the result / offset arrays live on the HEAP: a million result records are 64 MiB, and WITH-FOREIGN-OBJECTS puts
  ;; them on the control stack — 2 MiB on SBCL.  A member takes at least 20 octets, which bounds how many there can be.)
  (let* ((*alloc-results* nil) (members nil)
         (max-members (max 1 (min max-members (1+ (floor (- end start) 20)))))
         (res (cffi:foreign-alloc '(:struct tbz-result) :count max-members))
         (offs (cffi:foreign-alloc :uint64 :count max-members)))
    (cffi:with-foreign-objects ((n :size))
      (unwind-protect
           (progn
             (cffi:with-pointer-to-vector-data (pin compressed)
               (check-call (%inflate-gzip-members (engine) (cffi:inc-pointer pin start) (- end start)
                                                  (cffi:callback alloc-octets) (cffi:null-pointer) max-members res offs n)
                           "tbz_inflate_gzip_members"))
             (let ((cells (reverse *alloc-results*)))   ; one per delivered member, in order
               (dotimes (k (cffi:mem-ref n :size))
                 (let ((status (cffi:foreign-slot-value (cffi:mem-aptr res '(:struct tbz-result) k)
                                                        '(:struct tbz-result) 'status)))
                   (when (minusp status) (error "~a" (%strerror status)))
                   (unless (zerop status) (error "incomplete gzip stream"))
                   (push (take-alloc-result (pop cells)) members)))))
        (dolist (cell *alloc-results*)   ; (what was handed out and not taken — an error above — is given back)
          (unless (cffi:null-pointer-p (car cell)) (cffi:foreign-free (car cell))))
        (cffi:foreign-free res)
        (cffi:foreign-free offs)))
    (nreverse members)))

;;; ---- many independent streams over several devices (SURVEY §8e; north_star: "host code stays Common Lisp") ----
;;; One context per device; tbz_inflate_batch_multi assigns the streams (longest compressed first to the least loaded
;;; context), decodes each context's share in one batch call on a host thread of its own, and returns the results in
;;; stream order.  No collective: a deflate-state is self-contained (deflate.lisp:4-62).
(defvar *engines* nil "list of tbz_ctx pointers, one per device (OPEN-ENGINES)")
(defun open-engines (&optional (n (%device-count)))
  (setf *engines*
        (loop for d below n
              collect (cffi:with-foreign-object (p :pointer)
                        (let ((r (%ctx-create d p)))
                          (unless (zerop r) (error "tbz_ctx_create(~d): ~a" d (%strerror r)))
                          (cffi:mem-ref p :pointer))))))
(defun close-engines ()
  (mapc #'%ctx-destroy *engines*)
  (setf *engines* nil))

(defun decompress-vector-over-engines (compressed output &key (format :zlib) (start 0) (end (length compressed)))
  "ONE stream decoded by all the engines of OPEN-ENGINES together where it is a clean chain of flush-delimited parts
(tbz_inflate_sharded_multi: plan, a host thread per device, the seam proof), by the first engine alone otherwise.
Returns (values OUTPUT COUNT SHARDED-P).  Signals what DECOMPRESS-VECTOR signals."
  (let* ((engines (or *engines* (open-engines))) (k (length engines)))
    (cffi:with-foreign-objects ((ctxs :pointer k) (res '(:struct tbz-result)) (sh :int))
      (loop for e in engines for i from 0 do (setf (cffi:mem-aref ctxs :pointer i) e))
      (cffi:with-pointer-to-vector-data (pin compressed)
        (cffi:with-pointer-to-vector-data (pout output)
          (check-call (%inflate-sharded-multi ctxs k (format-code format) (cffi:inc-pointer pin start) (- end start)
                                              pout (length output) res sh)
                      "tbz_inflate_sharded_multi")))
      (cffi:with-foreign-slots ((status out-len) res (:struct tbz-result))
        (when (minusp status) (error "~a" (%strerror status)))
        (unless (zerop status)
          (if (= status 1)
              (error "incomplete ~a stream" format)
              (error "not enough space to decompress ~a stream" format)))
        (values output out-len (= 1 (cffi:mem-ref sh :int)))))))

(defun decompress-vectors (vectors outputs &key (format :zlib))
  "VECTORS: a list of octet vectors, each one stream.  OUTPUTS: a list of octet vectors to decode them into (as
DECOMPRESS-VECTOR's :OUTPUT).  Returns the list of counts.  A stream that fails signals what DECOMPRESS-VECTOR signals."
  (let* ((engines (or *engines* (open-engines)))
         (n (length vectors)) (k (length engines)) (fmt (format-code format)))
    (cffi:with-foreign-objects ((ctxs :pointer k) (ins :pointer n) (in-lens :size n) (outs :pointer n) (out-caps :size n)
                                (res '(:struct tbz-result) n))
      (loop for e in engines for i from 0 do (setf (cffi:mem-aref ctxs :pointer i) e))
      ;; every vector stays pinned for the whole call: nested WITH-POINTER-TO-VECTOR-DATA, innermost = the call
      (labels ((pin (vs os i)
                 (if (null vs)
                     (check-call (%inflate-batch-multi ctxs k fmt n ins in-lens outs out-caps res) "tbz_inflate_batch_multi")
                     (cffi:with-pointer-to-vector-data (pi_ (first vs))
                       (cffi:with-pointer-to-vector-data (po (first os))
                         (setf (cffi:mem-aref ins :pointer i) pi_
                               (cffi:mem-aref in-lens :size i) (length (first vs))
                               (cffi:mem-aref outs :pointer i) po
                               (cffi:mem-aref out-caps :size i) (length (first os)))
                         (pin (rest vs) (rest os) (1+ i)))))))
        (pin vectors outputs 0))
      (loop for i below n
            collect (cffi:with-foreign-slots ((status out-len) (cffi:mem-aptr res '(:struct tbz-result) i)
                                              (:struct tbz-result))
                      (when (minusp status) (error "stream ~d: ~a" i (%strerror status)))
                      (unless (zerop status)
                        (if (= status 1)
                            (error "stream ~d: incomplete ~a stream" i format)
                            (error "stream ~d: not enough space to decompress ~a stream" i format)))
                      out-len)))))
