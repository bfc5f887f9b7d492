// qds.h -- the subset of Qt's QDataStream wire format (version Qt_5_2, big-endian)
// that the reference's .fsim files and socket protocol use:
//   int / uint      4 bytes big-endian            quint64  8 bytes big-endian
//   float / double  8-byte IEEE double (QDataStream default DoublePrecision,
//                   so `qds << float` also writes 8 bytes: gpusim.cpp:409, :441)
//   char*           u32 length INCLUDING the NUL, bytes, NUL
//   QByteArray      u32 length, bytes (0xFFFFFFFF = null array)
// Written from the format description; no Qt involved.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace gpusim
{

class QdsReader
{
  public:
    QdsReader(const unsigned char* data, size_t size) : m_begin(data), m_p(data), m_end(data + size) {}
    explicit QdsReader(const std::vector<unsigned char>& v) : QdsReader(v.data(), v.size()) {}

    bool atEnd() const { return m_p >= m_end; }
    size_t remaining() const { return static_cast<size_t>(m_end - m_p); }
    size_t consumed() const { return static_cast<size_t>(m_p - m_begin); }

    uint32_t u32()
    {
        need(4);
        uint32_t v = (uint32_t(m_p[0]) << 24) | (uint32_t(m_p[1]) << 16) | (uint32_t(m_p[2]) << 8) | uint32_t(m_p[3]);
        m_p += 4;
        return v;
    }
    int32_t i32() { return static_cast<int32_t>(u32()); }
    uint64_t u64()
    {
        const uint64_t hi = u32();
        return (hi << 32) | u32();
    }
    double f64()
    {
        const uint64_t b = u64();
        double d;
        std::memcpy(&d, &b, 8);
        return d;
    }
    // char* string: returns it without the trailing NUL
    std::string cstr()
    {
        const uint32_t n = u32();
        need(n);
        std::string s(reinterpret_cast<const char*>(m_p), n ? n - 1 : 0);
        m_p += n;
        return s;
    }
    // a new[]-allocated copy, as `QDataStream >> char*&` hands out (gpusim.cpp:78-82)
    char* cstr_new()
    {
        const uint32_t n = u32();
        need(n);
        char* s = new char[n ? n : 1];
        if (n) std::memcpy(s, m_p, n);
        s[n ? n - 1 : 0] = '\0';
        m_p += n;
        return s;
    }
    void skip(size_t n)
    {
        need(n);
        m_p += n;
    }
    std::vector<unsigned char> bytearray()
    {
        const uint32_t n = u32();
        if (n == 0xFFFFFFFFu) return {};
        need(n);
        std::vector<unsigned char> v(m_p, m_p + n);
        m_p += n;
        return v;
    }

  private:
    void need(size_t n) const
    {
        if (static_cast<size_t>(m_end - m_p) < n) throw std::runtime_error("QDataStream: truncated input");
    }
    const unsigned char* m_begin;
    const unsigned char* m_p;
    const unsigned char* m_end;
};

class QdsWriter
{
  public:
    void u32(uint32_t v)
    {
        const unsigned char b[4] = {static_cast<unsigned char>(v >> 24), static_cast<unsigned char>(v >> 16),
                                    static_cast<unsigned char>(v >> 8), static_cast<unsigned char>(v)};
        m_buf.insert(m_buf.end(), b, b + 4);
    }
    void i32(int32_t v) { u32(static_cast<uint32_t>(v)); }
    void u64(uint64_t v)
    {
        u32(static_cast<uint32_t>(v >> 32));
        u32(static_cast<uint32_t>(v));
    }
    void f64(double d)
    {
        uint64_t b;
        std::memcpy(&b, &d, 8);
        u64(b);
    }
    void cstr(const char* s)
    {
        const size_t n = std::strlen(s) + 1;
        u32(static_cast<uint32_t>(n));
        m_buf.insert(m_buf.end(), s, s + n);
    }
    void bytearray(const unsigned char* p, size_t n)
    {
        u32(static_cast<uint32_t>(n));
        m_buf.insert(m_buf.end(), p, p + n);
    }
    const std::vector<unsigned char>& bytes() const { return m_buf; }
    std::vector<unsigned char>& bytes() { return m_buf; }

  private:
    std::vector<unsigned char> m_buf;
};

} // namespace gpusim
