// inputs.cpp -- see inputs.hpp.
#include "inputs.hpp"

#include <cerrno>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace vkmr {

// The bulk form on a stream that cannot be mapped (a pipe, a terminal, a socket).  A thread reads blocks of whole lines
// into three buffers in turn; GetBlock hands them out in order.  The state is shared (not owned by the Input) because a
// caller that stops early leaves the thread inside read(2): the Input then lets go of it instead of waiting.
struct Input::Reader {
    struct Block { std::vector<char> data; size_t len = 0; bool final = false; };
    Block blocks[3];
    std::deque<Block*> free, ready;
    Block* current = nullptr;   // the block the caller is looking at
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    int error = 0;              // errno of the read that failed, if one did (read under mu)

    // fd: this thread's OWN descriptor (a dup of the stream's): the Input may close its FILE while the thread sits in read(2),
    // and a number that is closed can be handed to the next open() of the process.  Closed here, when the thread ends.
    static void Run(std::shared_ptr<Reader> self, int fd, std::vector<char> tail)
    {
        struct Closer { int fd; ~Closer() { if (fd >= 0) close(fd); } } closer{fd};
        static const size_t kBlock = (size_t)16 << 20, kEnough = (size_t)8 << 20;
        for (;;) {
            Block* b = nullptr;
            {
                std::unique_lock<std::mutex> lock(self->mu);
                self->cv.wait(lock, [&] { return self->stop || !self->free.empty(); });
                if (self->stop) return;
                b = self->free.front();
                self->free.pop_front();
            }
            if (b->data.size() < kBlock) b->data.resize(kBlock);
            while (b->data.size() < tail.size() + kEnough) b->data.resize(b->data.size() * 2);
            if (!tail.empty()) memcpy(b->data.data(), tail.data(), tail.size());   // the unfinished line the last block ended with
            size_t end = tail.size();
            tail.clear();
            bool eof = false;
            size_t cut = 0, searched = 0;
            for (;;) {
                if (end == b->data.size()) b->data.resize(b->data.size() * 2);   // a line longer than the block
                const ssize_t got = read(fd, b->data.data() + end, b->data.size() - end);
                const int err = got < 0 ? errno : 0;   // before anything below can change it
                if (got < 0 && err == EINTR) continue;
                if (got < 0 && (err == EAGAIN || err == EWOULDBLOCK)) {   // a non-blocking stdin: wait for it, this is not the end
                    struct pollfd pfd = {fd, POLLIN, 0};
                    (void)poll(&pfd, 1, 1000);
                    std::lock_guard<std::mutex> lock(self->mu);
                    if (self->stop) return;
                    continue;
                }
                if (got <= 0) {
                    if (got < 0) {   // EIO, EBADF ...: the stream ends here, and the caller is told why
                        std::lock_guard<std::mutex> lock(self->mu);
                        self->error = err ? err : EIO;
                    }
                    eof = true;
                    break;
                }
                end += (size_t)got;
                if (end < kEnough) continue;   // a slow writer: several reads make a block worth a fork-join of the packer
                // the last '\n' of what has not been looked at yet (a line of gigabytes is scanned once, not once per read)
                const void* nl = memrchr(b->data.data() + searched, '\n', end - searched);
                searched = end;
                if (nl) {
                    cut = (size_t)(static_cast<const char*>(nl) - b->data.data()) + 1;
                    break;
                }
            }
            if (eof) {
                b->len = end;
                b->final = true;
            } else {
                b->len = cut;
                b->final = false;
                tail.assign(b->data.data() + cut, b->data.data() + end);
            }
            {
                std::lock_guard<std::mutex> lock(self->mu);
                self->ready.push_back(b);
            }
            self->cv.notify_all();
            if (eof) return;
        }
    }
};

Input::Input(FILE* fp, bool owner)
    : m_fp(fp), m_owner(owner), m_eof(fp == nullptr), m_size(0), m_count(0), m_buf(1 << 20), m_pos(0), m_end(0)
{
    struct stat st;
    if (fp && fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        const off_t at = lseek(fileno(fp), 0, SEEK_CUR);
        if (at == 0) {
            void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
            if (p != MAP_FAILED) {
                madvise(p, (size_t)st.st_size, MADV_SEQUENTIAL);
                m_map = static_cast<const char*>(p);
                m_map_len = (size_t)st.st_size;
            }
        }
    }
#ifdef F_SETPIPE_SZ
    // a pipe of 1 MiB instead of 64 KiB: fewer hand-overs between the writer and this process (365-430 ms instead of 400-450 for
    // 2.1 GB from `cat`, whose own limit is 350: profiles/r03_frontend_pipe.txt)
    if (fp && !m_map && fstat(fileno(fp), &st) == 0 && S_ISFIFO(st.st_mode)) (void)fcntl(fileno(fp), F_SETPIPE_SZ, 1 << 20);
#endif
}

Input::Input(const std::string& path) : Input(fopen(path.c_str(), "r"), true) {}

Input::~Input()
{
    if (m_reader) {
        {
            std::lock_guard<std::mutex> lock(m_reader->mu);
            m_reader->stop = true;
        }
        m_reader->cv.notify_all();   // the thread ends at its next wait, or when its read(2) returns: it holds the state until then
    }
    if (m_map) munmap(const_cast<char*>(m_map), m_map_len);
    if (m_owner && m_fp) fclose(m_fp);
}

size_t Input::ReadSome(char* dst, size_t n)
{
    if (m_map) {
        const size_t left = m_map_len - m_map_pos;
        const size_t take = n < left ? n : left;
        memcpy(dst, m_map + m_map_pos, take);
        m_map_pos += take;
        return take;
    }
    if (!m_fp) return 0;
    for (;;) {   // read(2), not fread: what a pipe holds now is worth having now
        const ssize_t got = read(fileno(m_fp), dst, n);
        if (got >= 0) return (size_t)got;
        if (errno == EINTR) continue;
        if (errno == EAGAIN || errno == EWOULDBLOCK) {
            struct pollfd pfd = {fileno(m_fp), POLLIN, 0};
            (void)poll(&pfd, 1, 1000);
            continue;
        }
        m_error = errno ? errno : EIO;
        return 0;
    }
}

int Input::Error() const
{
    if (m_error) return m_error;
    if (m_reader) {
        std::lock_guard<std::mutex> lock(m_reader->mu);
        return m_reader->error;
    }
    return 0;
}

bool Input::Fill()
{
    m_pos = 0;
    m_end = ReadSome(m_buf.data(), m_buf.size());
    return m_end > 0;
}

bool Input::GetView(const char** p, size_t* n)
{
    m_carry.clear();
    bool carried = false;
    for (;;) {
        if (m_pos == m_end && !Fill()) {
            m_eof = true;   // the read that hit EOF ends the current (possibly empty) line
            break;
        }
        const char* base = m_buf.data() + m_pos;
        const char* nl = static_cast<const char*>(memchr(base, '\n', m_end - m_pos));
        if (nl) {
            const size_t len = (size_t)(nl - base);
            m_pos += len + 1;
            if (!carried) {
                *p = base;
                *n = len;
                m_size += len;
                m_count += len ? 1 : 0;
                return true;
            }
            m_carry.append(base, len);
            break;
        }
        m_carry.append(base, m_end - m_pos);   // line continues in the next block
        carried = true;
        m_pos = m_end;
    }
    *p = m_carry.data();
    *n = m_carry.size();
    m_size += m_carry.size();
    m_count += m_carry.empty() ? 0 : 1;
    return true;
}

bool Input::GetBlock(const char** p, size_t* n, bool* final)
{
    if (m_map && m_pos == m_end) {
        // mapped file, nothing buffered: hand out the next span of the mapping, cut after a '\n'
        static const size_t span = [] { const char* e = getenv("VKMR_INPUT_SPAN_MB"); const long v = e ? atol(e) : 32; return (size_t)(v < 1 ? 1 : v) << 20; }();
        const size_t left = m_map_len - m_map_pos;
        size_t take = left;
        if (left > span) {
            const void* nl = memrchr(m_map + m_map_pos, '\n', span);
            if (nl) {
                take = (size_t)(static_cast<const char*>(nl) - (m_map + m_map_pos)) + 1;
            } else {   // a line longer than the span: up to its end
                const void* fwd = memchr(m_map + m_map_pos + span, '\n', left - span);
                take = fwd ? (size_t)(static_cast<const char*>(fwd) - (m_map + m_map_pos)) + 1 : left;
            }
        }
        *p = m_map + m_map_pos;
        *n = take;
        m_map_pos += take;
        *final = (m_map_pos == m_map_len);
        if (*final) m_eof = true;
        return true;
    }
    if (m_map) {   // a mapped file read line by line so far: what the buffer still holds is a copy of the mapping just before m_map_pos
        m_map_pos -= m_end - m_pos;
        m_pos = m_end = 0;
        return GetBlock(p, n, final);
    }
    // anything else: blocks of whole lines from the reading thread (started here, with what the line-at-a-time calls
    // may have left in the buffer)
    if (!m_reader) {
        m_reader = std::make_shared<Reader>();
        for (auto& blk : m_reader->blocks) m_reader->free.push_back(&blk);
        std::vector<char> tail(m_buf.begin() + (long)m_pos, m_buf.begin() + (long)m_end);
        m_pos = m_end = 0;
        const int own = m_fp ? dup(fileno(m_fp)) : -1;
        if (m_fp && own < 0) m_error = errno ? errno : EMFILE;   // the thread then reads nothing: the stream ends at once, with this error
        std::thread(Reader::Run, m_reader, own, std::move(tail)).detach();
    }
    Reader& r = *m_reader;
    std::unique_lock<std::mutex> lock(r.mu);
    if (r.current) {   // the caller is done with the block it was given last
        r.free.push_back(r.current);
        r.current = nullptr;
        r.cv.notify_all();
    }
    r.cv.wait(lock, [&] { return !r.ready.empty(); });
    r.current = r.ready.front();
    r.ready.pop_front();
    *p = r.current->data.data();
    *n = r.current->len;
    *final = r.current->final;
    if (*final) m_eof = true;
    return true;
}

std::string Input::Get()
{
    const char* p = nullptr;
    size_t n = 0;
    GetView(&p, &n);
    return std::string(p, n);
}

}  // namespace vkmr
